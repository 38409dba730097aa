/* edlib.h — drop-in for the part of Martinsos/edlib's C API that Raven's overlap hot path uses, backed by the
 * MI355X engine (libraven_hip.so).  Same names, argument meaning, result layout and ownership rules as edlib's own
 * header, so the call sites compile and behave unchanged:
 *     RavenLib/src/construct.cc:190-199   identity filter of ResolveContainedReads
 *     RavenLib/src/construct.cc:407-416   identity filter of the second pass
 *     (also RavenLib/src/assemble.cc:271-277, RavenLib/src/graph_repr.cc:250,361, RavenTest/src/raven_test.cpp:39-44)
 * all of which are  edlibAlign(q, qlen, t, tlen, edlibDefaultAlignConfig())  ->  result.status / result.editDistance
 * ->  edlibFreeAlignResult(result).
 *
 * What runs where: the distance is computed on the GPU by the batched Myers kernel behind rvn_edit_distance_batch
 * (raven_hip.h).  edlibAlign is a blocking single-pair call, and Raven issues it from many pool threads at once
 * (construct.cc:167-212); calls that are in flight at the same time are COMBINED into one device batch (a combining
 * queue inside the library), so N threads cost one upload + one launch, not N.  A hot loop that owns all its pairs
 * up front should call rvn_edit_distance_batch on spans of the uploaded reads instead (no inflate, no upload).
 *
 * Supported: EDLIB_MODE_NW with EDLIB_TASK_DISTANCE (edlibDefaultAlignConfig), any k (k < 0 = unbounded; a distance
 * above k >= 0 is reported as -1 exactly as edlib does), sequences over at most 4 distinct symbols (the device
 * works on 2-bit codes; equality of bytes is what is compared, like edlib without additional equalities).
 * Anything else — SHW / HW modes, LOC / PATH tasks, additional equalities, more than 4 distinct symbols, no usable
 * GPU — returns status EDLIB_STATUS_ERROR; there is no CPU path in this library. */
#ifndef EDLIB_H
#define EDLIB_H

#ifdef __cplusplus
extern "C" {
#endif

#define EDLIB_STATUS_OK 0
#define EDLIB_STATUS_ERROR 1

typedef enum { EDLIB_MODE_NW, EDLIB_MODE_SHW, EDLIB_MODE_HW } EdlibAlignMode;
typedef enum { EDLIB_TASK_DISTANCE, EDLIB_TASK_LOC, EDLIB_TASK_PATH } EdlibAlignTask;
typedef enum { EDLIB_CIGAR_STANDARD, EDLIB_CIGAR_EXTENDED } EdlibCigarFormat;

#define EDLIB_EDOP_MATCH 0
#define EDLIB_EDOP_INSERT 1
#define EDLIB_EDOP_DELETE 2
#define EDLIB_EDOP_MISMATCH 3

typedef struct {
  char first;
  char second;
} EdlibEqualityPair;

typedef struct {
  int k;                /* >= 0: report -1 when the distance is larger; < 0: unbounded */
  EdlibAlignMode mode;  /* only EDLIB_MODE_NW */
  EdlibAlignTask task;  /* only EDLIB_TASK_DISTANCE */
  const EdlibEqualityPair* additionalEqualities; /* must be NULL */
  int additionalEqualitiesLength;                /* must be 0 */
} EdlibAlignConfig;

EdlibAlignConfig edlibNewAlignConfig(int k, EdlibAlignMode mode, EdlibAlignTask task,
                                     const EdlibEqualityPair* additionalEqualities, int additionalEqualitiesLength);
/* k = -1, mode = EDLIB_MODE_NW, task = EDLIB_TASK_DISTANCE, no additional equalities */
EdlibAlignConfig edlibDefaultAlignConfig(void);

typedef struct {
  int status;          /* EDLIB_STATUS_OK / EDLIB_STATUS_ERROR */
  int editDistance;    /* -1 when larger than k */
  int* endLocations;   /* NW: { targetLength - 1 }; NULL when editDistance == -1; freed by edlibFreeAlignResult */
  int* startLocations; /* NULL (distance task) */
  int numLocations;
  unsigned char* alignment; /* NULL (distance task) */
  int alignmentLength;
  int alphabetLength; /* distinct symbols in query and target */
} EdlibAlignResult;

void edlibFreeAlignResult(EdlibAlignResult result);

/* query / target need not be zero-terminated */
EdlibAlignResult edlibAlign(const char* query, int queryLength, const char* target, int targetLength,
                            const EdlibAlignConfig config);

/* edlib's CIGAR printer (host string formatting only; provided so that code using it links).  Returns a malloc'ed,
 * zero-terminated string the caller frees, or NULL. */
char* edlibAlignmentToCigar(const unsigned char* alignment, int alignmentLength, EdlibCigarFormat cigarFormat);

#ifdef __cplusplus
}
#endif
#endif /* EDLIB_H */
