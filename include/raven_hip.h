/* raven_hip.h — C ABI of the MI355X-native overlap-and-polish engine for Raven (libraven_hip.so).
 *
 * Drop-in boundary: the entry points below are what a binding for Raven's overlap and
 * polishing hot path would bind instead of the un-vendored ram::MinimizerEngine, edlib and racon::Polisher;
 * each cites the reference interface it replaces (paths relative to the lbcb-sci/raven tree).  Plain
 * pointers and sizes only.  Groups: engine / reads; Minimize / Filter / Map; FindOverlapsAndCreatePiles;
 * Pile::AddLayers / AddKmers; edit distance; POA window consensus; polishing round (whole, or a window range
 * for sharding); stage-level entry points of the sharded pass (host and device-pointer variants);
 * introspection and test hooks.
 *
 * Conventions
 *   - every function returning int returns RVN_OK (0) or a negative RVN_E* code; rvn_last_error()
 *     gives the thread-local message.  The C++ facade (include/ram/minimizer_engine.hpp) rethrows
 *     RVN_EINVAL as std::invalid_argument, matching ram/biosoup behaviour.
 *   - reads are handed over in biosoup::NucleicAcid layout: 2 bits per base, 32 bases per uint64_t,
 *     base i at bits (2i mod 64) of word i/32, A=0 C=1 G=2 T=3; every read starts on a word boundary
 *     (i.e. the concatenation of each read's `deflated_data`); `word_offsets` has n+1 entries.
 *   - rvn_overlap == biosoup::Overlap without the alignment string (8 x uint32_t).
 *   - the library fails loudly (RVN_ENODEVICE) when no HIP device is present; there is no CPU path.
 */
#ifndef RAVEN_HIP_H_
#define RAVEN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVN_OK 0
#define RVN_EINVAL (-1)    /* invalid argument (ram: std::invalid_argument) */
#define RVN_ENODEVICE (-2) /* no usable HIP device */
#define RVN_EHIP (-3)      /* HIP runtime / kernel failure */
#define RVN_ENOMEM (-4)

typedef struct rvn_engine rvn_engine; /* one per (device, k, w): replaces ram::MinimizerEngine */
typedef struct rvn_reads rvn_reads;   /* device-resident packed read set */
typedef struct rvn_pass1 rvn_pass1;   /* result of FindOverlapsAndCreatePiles, resident in HBM */

typedef struct rvn_overlap {
  uint32_t lhs_id, lhs_begin, lhs_end;
  uint32_t rhs_id, rhs_begin, rhs_end;
  uint32_t score;
  uint32_t strand;
} rvn_overlap;

const char* rvn_last_error(void);
int rvn_device_count(void);

/* ram::MinimizerEngine{thread_pool, k, w[, bandwidth=500, chain=4, matches=100, gap=10000]}
 * (RavenLib/src/construct.cc:661-662, RavenLib/src/assemble.cc:753).  k is clamped to [1,31] as in ram. */
int rvn_engine_create(rvn_engine** out, uint32_t k, uint32_t w, uint32_t bandwidth, uint32_t chain,
                      uint32_t matches, uint32_t gap, int device);
void rvn_engine_destroy(rvn_engine* e);

/* Copies the packed reads to HBM once; they stay resident for any number of calls
 * (the reference keeps std::vector<std::unique_ptr<biosoup::NucleicAcid>> in RAM, RavenExe/src/main.cc:258-299). */
int rvn_reads_upload(rvn_engine* e, const uint64_t* packed, uint64_t n_words, const uint64_t* word_offsets,
                     const uint32_t* lengths, const uint32_t* ids, uint32_t n_reads, rvn_reads** out);
/* Same from one-byte base codes (0..3; read i = codes[offsets[i] .. offsets[i+1])): the 2-bit packing happens on the
 * device.  This is how the consensus of one polishing round becomes the target set of the next
 * (RavenLib/src/polish.cc:50-71 rebuilds biosoup::NucleicAcid objects from the polished strings). */
int rvn_reads_upload_codes(rvn_engine* e, const uint8_t* codes, const uint64_t* offsets, const uint32_t* ids,
                           uint32_t n_reads, rvn_reads** out);

/* The consensus of the engine's last complete rvn_polish_round (every window: not a _range call over a part) as a read
 * set, straight from HBM.  Replaces, for rounds after the first, the upload of the targets raven::Polish hands the next
 * racon::Polisher (RavenLib/src/polish.cc:43-74: the polished sequences of round r are round r + 1's targets); the result
 * equals rvn_reads_upload_codes of the sequences the round returned, bit for bit.  RVN_EINVAL when no such consensus is
 * resident (no round yet, a partial round, or the engine released its scratch since). */
int rvn_polish_output_as_reads(rvn_engine* engine, rvn_reads** out);
void rvn_reads_destroy(rvn_reads* r);
/* The input path: raven::CreateParser(path) + Parse(-1) (RavenLib/src/io.cc:7-41, RavenExe/src/main.cc:258-299) straight
 * into HBM.  Format by extension exactly as io.cc (.fasta / .fa / .fastq / .fq, optionally .gz; anything else is
 * RVN_EINVAL with the reference's message).  A gzip file is cut into its members (BGZF blocks, concatenated members) and a
 * pool of host threads inflates them straight into page-locked slabs of the text (a single-member archive is one deflate
 * stream: one thread, front to back); the caller's thread finds the records with one memchr pass per slab, copies the
 * slab to HBM as it is, and the device cuts the records out and 2-bit packs them (biosoup's coder table: IUPAC folded,
 * any other character is RVN_EINVAL "not a nucleotide").  A damaged or truncated archive is RVN_EINVAL.  Read ids = 0 .. n-1 in file order.  FASTQ: biosoup's block qualities (integer mean of
 * every 64-base block) are computed on the device and attached to the read set (as rvn_reads_attach_quality with
 * block_shift 6 would).  rvn_reads_name: first word of the header of sequence i. */
typedef struct rvn_load_stats {
  uint64_t n_sequences, n_bases;
  int has_quality;
  double parse_s, device_s, total_s; /* record scanner | copies + packing on the device (caller thread) | whole call */
  uint32_t inflate_threads, members; /* the inflate pool (host threads) and the gzip members / file pieces it worked on */
  int32_t streaming, restarted; /* 1: one deflate stream, inflated front to back by one thread | 1: a member cut was wrong */
} rvn_load_stats;
int rvn_reads_load(rvn_engine* e, const char* path, rvn_reads** out, rvn_load_stats* stats);
const char* rvn_reads_name(const rvn_reads* r, uint32_t i);
int rvn_reads_info(const rvn_reads* r, uint32_t* n_reads, uint64_t* n_words, uint64_t* n_bases, uint64_t* n_quality_bytes,
                   int* quality_shift);
/* read-back of a device-resident read set (tests, callers that need the packed words on the host) */
int rvn_reads_fetch(const rvn_reads* r, uint64_t* packed, uint64_t* word_offsets, uint32_t* lengths, uint8_t* quals,
                    uint64_t* quality_offsets);
/* Base qualities of an uploaded read set, kept in HBM beside the bases for every later polishing round
 * (biosoup::NucleicAcid::block_quality; raven::Polish computes its threshold from them, polish.cc:26-41).
 * quals: Phred+33 bytes, one per 2^block_shift bases of a read (block_shift 0: per base; 6: biosoup's block qualities
 * = the mean of 64 bases + 33, which is all racon ever sees of them), read i at offsets[i] (offsets[n_reads+1]).
 * rvn_polish_round uses them when called with read_quals == NULL.  quals == NULL detaches. */
int rvn_reads_attach_quality(rvn_engine* e, rvn_reads* r, const uint8_t* quals, const uint64_t* offsets, int block_shift);

/* ram::MinimizerEngine::Minimize(first, last, minhash) — builds the index (construct.cc:42-43, :363). */
int rvn_engine_minimize(rvn_engine* e, const rvn_reads* r, uint32_t first, uint32_t last, int minhash);
/* ram::MinimizerEngine::Filter(frequency) (construct.cc:44, :372); RVN_EINVAL unless 0 <= f <= 1. */
int rvn_engine_filter(rvn_engine* e, double frequency);
uint32_t rvn_engine_occurrence(const rvn_engine* e);

/* Batched ram::MinimizerEngine::Map(sequence, avoid_equal, avoid_symmetric, minhash[, filtered])
 * for reads [first, last) (construct.cc:59-64 / :377-381).  Results stay on the device until fetched;
 * overlaps come back in (read, Map-output) order with per-read offsets (last-first+1 entries). */
int rvn_engine_map_batch(rvn_engine* e, const rvn_reads* r, uint32_t first, uint32_t last, int avoid_equal,
                         int avoid_symmetric, int minhash, int want_filtered, uint64_t* n_overlaps);
int rvn_engine_map_fetch(rvn_engine* e, rvn_overlap* overlaps, uint32_t* read_offsets);
/* positions of query minimizers skipped by the occurrence filter (`filtered` argument of Map),
 * per read: offsets (last-first+1) into `positions`.  Pass positions == NULL to query the total. */
int rvn_engine_map_fetch_filtered(rvn_engine* e, uint32_t* positions, uint32_t* read_offsets, uint64_t* total);
/* Map + fetch as ONE critical section: ram::MinimizerEngine::Map is const and Raven calls it concurrently from its
 * pool workers (construct.cc:60-64, :373-381), so a caller that cannot hold its own lock across
 * rvn_engine_map_batch / rvn_engine_map_fetch uses this.  Every entry point of this library takes the engine's lock,
 * so concurrent calls on one engine are safe; this one additionally returns the results of ITS map (not a later
 * caller's).  Outputs are malloc'ed by the library (overlaps of all reads concatenated, read_offsets[last-first+1];
 * with want_filtered also the `filtered` positions and their offsets) and released with rvn_free. */
int rvn_engine_map_collect(rvn_engine* e, const rvn_reads* r, uint32_t first, uint32_t last, int avoid_equal,
                           int avoid_symmetric, int minhash, int want_filtered, rvn_overlap** overlaps,
                           uint32_t** read_offsets, uint32_t** filtered, uint32_t** filtered_offsets);
void rvn_free(void* p);
/* Hands the engine's scratch and intermediate buffers (index, sketches, last Map result, stage scratch) back to the
 * allocator.  They only ever grow, so that steady-state calls never touch the allocator; the stage entry points
 * (first pass, second pass, polishing round) call this themselves when less than a third of the HBM is free.
 * Result handles (rvn_reads, rvn_pass1, rvn_pass2) are not affected; rvn_engine_map_fetch results are. */
int rvn_engine_release_scratch(rvn_engine* e);

/* raven::FindOverlapsAndCreatePiles (RavenLib/src/construct.cc:14-121; decl construct.h:22-29):
 * index batches of `index_batch_bases` (reference: 1<<32), query flushes of `flush_bases` (reference:
 * 1<<30), merge, Pile::AddLayers (pile.cc:33-62) and the top-kMax truncation, all on the device.
 * Requires ids[i] == i (the reference's own invariant, construct.cc:25,74-75). */
int rvn_find_overlaps_and_create_piles(rvn_engine* e, const rvn_reads* r, double freq, uint32_t k_max_overlaps,
                                       int use_minhash, uint64_t index_batch_bases, uint64_t flush_bases,
                                       rvn_pass1** out);
uint64_t rvn_pass1_pile_words(const rvn_pass1* p);   /* sum over reads of (len >> 4) */
uint64_t rvn_pass1_num_overlaps(const rvn_pass1* p); /* sum of per-pile list sizes */
/* pile coverage: uint16 data of all piles concatenated + offsets[n+1] (Pile::data_, pile.h:133) */
int rvn_pass1_fetch_piles(const rvn_pass1* p, uint16_t* data, uint64_t* offsets);
/* overlaps[i] of construct.cc:666, concatenated + offsets[n+1] */
/* raven::TrimAndAnnotatePiles' first two steps on the coverage arrays where they are, in HBM
 * (RavenLib/src/construct.cc:131-139): Pile::FindValidRegion(coverage) + UpdateValidRegion + FindMedian
 * (pile.cc:122-174) for every pile.  begin / end in cells (Pile::begin_ / end_, i.e. bases >> 4), invalid = the
 * pile's is_invalid flag (the caller clears overlaps[i] for those, construct.cc:134-135); the coverage of valid piles
 * is zeroed outside the region exactly as UpdateValidRegion does (visible through rvn_pass1_fetch_piles).
 * FindChimericRegions (double-precision slopes) stays on the host. */
int rvn_pass1_trim_and_annotate(rvn_pass1* p, uint32_t coverage, uint32_t* begin, uint32_t* end, uint16_t* median,
                                uint8_t* invalid);
/* The third step of raven::TrimAndAnnotatePiles (construct.cc:139): Pile::FindChimericRegions (pile.cc:176-187) =
 * FindSlopes(1.82) (pile.cc:403-600: coverage drops / rises against the sliding maxima within 52 cells, evaluated in
 * double as the reference does), down-slope / up-slope pairing and MergeRegions (pile.cc:373-400), for every pile that
 * rvn_pass1_trim_and_annotate left valid (`invalid` = its output), on the coverage arrays in HBM.  Output: the piles'
 * chimeric_regions_ as (begin, end) cell pairs, pile i at region_offsets[i] .. region_offsets[i+1] (pairs; n + 1 entries,
 * caller's array); *regions is malloc'ed by the library (2 uint32 per pair), release with rvn_free.
 * Pile::is_maybe_chimeric() == (region_offsets[i+1] > region_offsets[i]). */
int rvn_pass1_find_chimeric_regions(rvn_pass1* p, const uint8_t* invalid, uint32_t* region_offsets, uint32_t** regions);
int rvn_pass1_fetch_overlaps(const rvn_pass1* p, rvn_overlap* overlaps, uint32_t* offsets);
void rvn_pass1_destroy(rvn_pass1* p);

/* raven::FindOverlapsAndRepetetiveRegions (RavenLib/src/construct.cc:316-491; decl construct.h:49-54), the second
 * all-vs-all pass on the VALID reads: valid reads first by id, index batches of `batch_bases` (reference: 1 << 30)
 * without minhash, Map(read, true, true, false, &filtered) + Pile::AddKmers(filtered, kmer_len) (pile.cc:64-120),
 * the identity filter when identity != 0 (construct.cc:385-424: OverlapUpdate, edlibAlign of the two spans, drop below
 * the threshold), then the merge of construct.cc:430-455 — OverlapUpdate, GetOverlapType (overlap_utils.cc:14-113),
 * containment flags, consecutive overlaps of the same pair keep the longer — the contained piles turned invalid and
 * the final OverlapUpdate sweep (construct.cc:466-480), all on the device.
 *   r               ALL reads, ids[i] == i
 *   pile_begin/end  Pile::begin() / Pile::end() of every pile in BASES (begin_ << 4), pile_invalid = Pile::is_invalid()
 * Result (rvn_pass2_fetch): the overlaps the reference leaves in the extra slot overlaps.back(), contained[n] = piles
 * this pass marked with set_is_contained() (the caller also sets them invalid, construct.cc:466-470), and the k-mer
 * cells Pile::kmers_ of the valid reads: (len >> 4) + 1 bytes (0/1) per valid read at kmers_offsets[id], nothing for
 * invalid reads (kmers_offsets has n + 1 entries; rvn_pass2_kmer_cells = their total). */
typedef struct rvn_pass2 rvn_pass2;
int rvn_find_overlaps_and_repetitive_regions(rvn_engine* e, const rvn_reads* r, const uint32_t* pile_begin,
                                             const uint32_t* pile_end, const uint8_t* pile_invalid, double freq,
                                             uint32_t kmer_len, double identity, uint64_t batch_bases, rvn_pass2** out);
uint64_t rvn_pass2_num_overlaps(const rvn_pass2* p);
uint64_t rvn_pass2_kmer_cells(const rvn_pass2* p);
int rvn_pass2_fetch(const rvn_pass2* p, rvn_overlap* overlaps, uint8_t* contained, uint8_t* kmers, uint64_t* kmers_offsets);
void rvn_pass2_destroy(rvn_pass2* p);

/* The identity filter loop of raven::ResolveContainedReads (RavenLib/src/construct.cc:162-217) on the per-pile overlap
 * lists overlaps[i] (concatenated, offsets[n+1], both updated in place): every overlap goes through OverlapUpdate
 * (overlap_utils.cc:14-80; dropped when it fails), its two spans through the batched exact edit distance (rhs
 * reverse-complemented on the opposite strand), and is kept with its updated coordinates when
 * 1 - distance / max(length) >= identity.  The containment marking that follows in the reference is host graph logic. */
int rvn_filter_overlaps_by_identity(rvn_engine* e, const rvn_reads* r, rvn_overlap* overlaps, uint32_t* offsets,
                                    const uint32_t* pile_begin, const uint32_t* pile_end, const uint8_t* pile_invalid,
                                    double identity);

/* raven::Pile::AddLayers on one pile (RavenLib/src/pile.cc:33-62): `data` (cells = len >> 4) is updated
 * in place with the coverage of `n` overlaps touching pile `id`. */
int rvn_pile_add_layers(rvn_engine* e, uint16_t* data, uint32_t cells, uint32_t id, const rvn_overlap* overlaps,
                        uint64_t n);

/* raven::Pile::AddKmers (RavenLib/src/pile.cc:64-120; call site construct.cc:382) for reads
 * [first_read, first_read + n_reads): `positions` are the `filtered` outputs of Map (rvn_engine_map_fetch_filtered)
 * concatenated with position_offsets[n_reads+1]; `kmers` holds, per read, Pile::kmers_ as (len >> 4) + 1 bytes
 * (0/1) at kmers_offsets[i]; cells of k-mers that pass the low-complexity filter are set to 1. */
int rvn_pile_add_kmers_batch(rvn_engine* e, const rvn_reads* r, uint32_t first_read, uint32_t n_reads,
                             const uint32_t* positions, const uint64_t* position_offsets, uint8_t* kmers,
                             const uint64_t* kmers_offsets);

/* Batched edlibAlign(lhs, rhs, edlibDefaultAlignConfig()).editDistance (global / NW, unit costs) between
 * spans of uploaded reads: RavenLib/src/construct.cc:176-199 (identity filter of ResolveContainedReads) and
 * :393-416 (second pass).  lhs/rhs_read are read INDICES in `r`; strand == 0 reverse-complements the rhs span
 * first (construct.cc:184-188).  distances[i] is the exact edit distance (any size; no threshold). */
typedef struct rvn_ed_pair {
  uint32_t lhs_read, lhs_begin, lhs_len;
  uint32_t rhs_read, rhs_begin, rhs_len;
  uint32_t strand;
  uint32_t reserved;
} rvn_ed_pair;
int rvn_edit_distance_batch(rvn_engine* e, const rvn_reads* r, const rvn_ed_pair* pairs, uint32_t n_pairs,
                            uint32_t* distances, double* device_ms, uint64_t* cells);

/* Batched racon Window::GenerateConsensus (the POA consensus behind racon::Polisher::Polish,
 * RavenLib/src/polish.cc:43-51; scores = AlignCfg of polish.hpp:13-17): for every window, layer 0 is the
 * backbone, the others are the read pieces aligned to it.
 *   codes            base codes 0..3 of all layers of all windows, concatenated
 *   quals            Phred+33 characters parallel to codes, or NULL; has_qual[layer] selects per layer
 *                    (racon gives a backbone without quality a dummy '!' string, i.e. weight 0)
 *   layer_offsets    [n_layers+1] into codes;  begins/ends [n_layers]: first/last covered backbone position
 *   window_offsets   [n_windows+1] into the layer tables
 *   consensus        output codes, window w at consensus_offsets[w] (capacity = next offset - this one)
 *   status           0: fewer than 3 sequences, backbone returned (racon: unpolished)   1: polished
 *                    2: window exceeds the device limits (nodes / layer length / in-degree), backbone returned
 * Results agree with the CPU path within edit-distance tolerance (ties between equal-score paths may resolve
 * differently). */
int rvn_poa_consensus_batch(rvn_engine* e, const uint8_t* codes, const uint8_t* quals, const uint64_t* layer_offsets,
                            const uint32_t* begins, const uint32_t* ends, const uint32_t* has_qual,
                            const uint32_t* window_offsets, uint32_t n_windows, int match, int mismatch, int gap,
                            int trim, uint8_t* consensus, const uint64_t* consensus_offsets, uint32_t* consensus_len,
                            uint32_t* status, double* device_ms);

/* One polishing round == racon::Polisher::Polish(targets, sequences, drop_unpolished = false) as raven::Polish calls
 * it (RavenLib/src/polish.cc:43-51: e = 0.3, w = 500, trim = true; `q` = the average read quality computed at
 * polish.cc:26-41; match/mismatch/gap = AlignCfg).  `e` must have been created with k = 15, w = 5 (racon's own
 * minimizer engine).  targets / reads are uploaded read sets; read_quals = per-base Phred+33 of every read
 * concatenated in read order (qual_offsets[n_reads+1]) or NULL.  Output: polished base codes of target t at
 * out_offsets[t] (capacity out_offsets[t+1]-out_offsets[t]; 2 x length + 1024 is ample), out_len[t], and the
 * polished-window ratio racon writes into the XC:f: tag (polish.cc:57-59 tests it for > 0).
 * Window breakpoints come from the exact global alignment path of every read's best overlap, as in racon
 * (DESIGN.md §3.7); read_quals == NULL uses the qualities attached with rvn_reads_attach_quality, if any. */
typedef struct rvn_polish_stats {
  uint64_t n_overlaps, n_reads_used, n_layers, n_windows, n_polished_windows, n_failed_windows;
  double poa_ms;                    /* device time of the window-consensus batch */
  double map_ms, host_ms, total_ms; /* wall: index + map + best overlap | host planning (jobs, window tables) | whole call */
  uint64_t n_dropped_layers;        /* reads left out because their alignment is beyond the path kernel (band > ~32 000) */
  double align_ms;                  /* device time of the alignment-path stage (banded NW forward + traceback) */
  uint64_t n_aligned, n_align_retries; /* read-to-target alignments done | attempts repeated with a doubled band */
  uint64_t align_band_cells;        /* DP cells inside the computed bands (all attempts) */
  uint64_t align_store_bytes;       /* largest band store of one batch (16 + 4 bytes per 64 cells) */
} rvn_polish_stats;
int rvn_polish_round(rvn_engine* e, rvn_reads* targets, rvn_reads* reads, const uint8_t* read_quals,
                     const uint64_t* qual_offsets, double q, double err, uint32_t w, int trim, int match, int mismatch,
                     int gap, uint8_t* out_codes, const uint64_t* out_offsets, uint32_t* out_len, double* ratio,
                     rvn_polish_stats* stats);

/* ---- stage-level entry points of the SHARDED single-genome pass (one engine per GPU) ------------
 * FindOverlapsAndCreatePiles (construct.cc:14-121) split where the data has to move between GPUs.  Reads are
 * range-partitioned by pile (rank g owns a contiguous id range and uploads only those reads, ids = global read
 * indices), minimizer values are partitioned by hash class.  Host side: raven_amd/sharded.py.
 *   1. rvn_shard_sketch           own reads -> (value, origin) with the minhash-selected entries flagged (bit 63 of
 *                                 origin) == the per-read part of ram Minimize + the query sketch of Map
 *      -- all-to-all #1: every minimizer to the owner of its hash class --
 *   2. rvn_shard_index_build      owner: stable sort by value == its hash class of ram's index, same relative order
 *   3. rvn_shard_key_counts       per-key counts -> all-reduce of the count histogram -> exact global Filter
 *      rvn_engine_set_occurrence
 *   4. rvn_shard_join             owner: index probes of Map for every query entry of its hash class; matches
 *                                 segmented by the query read's GLOBAL id
 *      -- all-to-all #2: every match to the owner of its query read (the candidate-pair exchange) --
 *   5. rvn_shard_chain            read owner: sort / bands / LIS / emission of Map -> rvn_engine_map_fetch
 *      -- all-to-all #3: every overlap also to the owner of its rhs read --
 *   6. rvn_shard_piles            pile owner: merge (construct.cc:72-77), AddLayers, top-kMax truncation
 * The result for the reads a rank owns is bit-identical to the single-GPU pass (tests/test_gpu_sharded.py). */
int rvn_shard_sketch(rvn_engine* e, const rvn_reads* own_reads, int index_minhash, uint64_t* count);
/* The same for the reads [first, last) of `own` (indices into the handle).  foreign != 0: these reads belong to an EARLIER
 * index batch of a pass with several (construct.cc:32-37: an index batch is 2^32 bases; every read up to the batch's end is
 * mapped against it, :59-64): only their minhash-selected minimizers come back, flagged query-only (bit 62 of the origin
 * word beside the query flag, bit 63) — the index shard sorts them in beside its members, the self-join matches them
 * against the members and never counts them towards a key's occurrence.  Fetch with rvn_shard_sketch_fetch[_dev] before
 * the next call. */
int rvn_shard_sketch_range(rvn_engine* e, const rvn_reads* own, uint32_t first, uint32_t last, int index_minhash, int foreign,
                           uint64_t* count);
int rvn_shard_sketch_fetch(rvn_engine* e, uint64_t* values, uint64_t* origins);
int rvn_shard_index_build(rvn_engine* e, const uint64_t* values, const uint64_t* origins, uint64_t n, int all_query);
int rvn_shard_key_counts(rvn_engine* e, uint32_t* counts /* n_keys of rvn_engine_index_size */);
int rvn_engine_set_occurrence(rvn_engine* e, uint32_t occurrence);
int rvn_shard_join(rvn_engine* e, uint32_t n_reads_total, int avoid_equal, int avoid_symmetric, uint64_t* n_matches);
/* same, restricted to the query reads [query_first, query_last) of one flush window (construct.cc:56-70): a pass with
 * several flush windows joins, chains and merges window by window, exactly as the reference flushes */
int rvn_shard_join_range(rvn_engine* e, uint32_t n_reads_total, int avoid_equal, int avoid_symmetric,
                         uint32_t query_first, uint32_t query_last, uint64_t* n_matches);
int rvn_shard_join_fetch(rvn_engine* e, uint64_t* group, uint64_t* positions, uint64_t* seg_off /* n_reads_total+1 */);
int rvn_shard_chain(rvn_engine* e, const rvn_reads* own_reads, const uint64_t* group, const uint64_t* positions,
                    const uint64_t* seg_off /* own n + 1 */, uint64_t* n_overlaps);
/* overlaps: Map outputs in (query read, emission) order over ALL reads' id space; only piles of reads whose
 * overlaps are complete in the list (the caller's own range) are meaningful in the returned handle */
int rvn_shard_piles(rvn_engine* e, const uint32_t* lengths, uint32_t n_reads_total, const rvn_overlap* overlaps,
                    uint64_t n, uint32_t kmax, rvn_pass1** out);

/* piles kept across the flush windows of a pass: create once, merge the Map outputs of every window (each merge =
 * the serial merge + AddLayers + top-kMax truncation of one flush, construct.cc:72-110); rvn_shard_piles = both */
int rvn_shard_piles_create(rvn_engine* e, const uint32_t* lengths, uint32_t n_reads_total, rvn_pass1** out);
int rvn_shard_piles_merge(rvn_pass1* p, const rvn_overlap* overlaps, uint64_t n, uint32_t kmax);
int rvn_shard_piles_merge_dev(rvn_pass1* p, const rvn_overlap* d_overlaps, const uint32_t* d_overlap_read_off, uint64_t n,
                              uint32_t kmax);

/* Device-pointer variants of the same stages: every d_* argument is a pointer into HBM owned by the caller (the
 * torch CUDA tensors the exchanges run on), so nothing crosses PCIe between the stages.  Calls are synchronous with
 * respect to the engine's stream; the caller synchronises its own stream before passing buffers in. */
int rvn_shard_sketch_fetch_dev(rvn_engine* e, uint64_t* d_values, uint64_t* d_origins);
int rvn_shard_index_build_dev(rvn_engine* e, const uint64_t* d_values, const uint64_t* d_origins, uint64_t n,
                              int all_query, uint64_t n_flagged /* origins with bit 63 set */);
/* count-of-counts of this shard's keys: hist[65536] (host; bin 65535 = number of keys with count >= 65535, whose
 * counts go to over[0..*n_over)) */
int rvn_shard_key_histogram(rvn_engine* e, uint64_t* hist, uint32_t* over, uint32_t over_cap, uint32_t* n_over);
int rvn_shard_join_fetch_dev(rvn_engine* e, uint64_t* d_group, uint64_t* d_positions, uint64_t* d_seg_off);
int rvn_shard_chain_dev(rvn_engine* e, const rvn_reads* own_reads, const uint64_t* d_group, const uint64_t* d_positions,
                        const uint64_t* d_seg_off, uint64_t n_matches, uint64_t* n_overlaps);
int rvn_engine_map_fetch_dev(rvn_engine* e, rvn_overlap* d_overlaps, uint32_t* d_read_offsets);
/* Partition / regroup steps between the stages (device pointers; `counts`, `bounds`, `n_per_source` and the pointer tables
 * themselves are host memory).  They replace what a host would do with sort / bincount / gather:
 *   split_minimizers  stable partition of (value, origin) by owner rank = hash class of the value (the order inside a
 *                     rank's part is the input order = ram's (read, position) order); counts[world]
 *   count_flagged     number of origins with bit 63 set (the minhash-selected query entries) in a received buffer
 *   adjacent_diff     per-read match counts out of the per-read offsets of rvn_shard_join_fetch_dev
 *   regroup           matches received from every index owner -> per-read segments, sources in rank order
 *   split_overlaps    stable partition of Map's overlaps by the owner of their rhs read (bounds[world + 1] = read ranges);
 *                     counts[world + 1], the last bucket = overlaps whose rhs read is `self`'s (they stay and are not copied
 *                     to d_out's first buckets)
 *   merge_parts       rvn_shard_piles_merge_dev on the concatenation of `n_parts` lists (received parts in rank order, own
 *                     overlaps last), per-read offsets computed on the device */
int rvn_shard_split_minimizers_dev(rvn_engine* e, const uint64_t* d_values, const uint64_t* d_origins, uint64_t n,
                                   uint32_t world, uint64_t* d_values_out, uint64_t* d_origins_out, uint64_t* counts);
int rvn_shard_count_flagged_dev(rvn_engine* e, const uint64_t* d_origins, uint64_t n, uint64_t* count);
int rvn_shard_adjacent_diff_dev(rvn_engine* e, const uint64_t* d_seg_off, uint64_t n, uint64_t* d_counts);
int rvn_shard_regroup_dev(rvn_engine* e, uint32_t world, const uint64_t* const* d_counts, const uint64_t* const* d_group,
                          const uint64_t* const* d_positions, const uint64_t* n_per_source, uint32_t n_reads,
                          uint64_t* d_seg_off, uint64_t* d_group_out, uint64_t* d_positions_out);
int rvn_shard_split_overlaps_dev(rvn_engine* e, const rvn_overlap* d_overlaps, uint64_t n, const uint32_t* bounds,
                                 uint32_t world, uint32_t self, rvn_overlap* d_out, uint64_t* counts);
int rvn_shard_piles_merge_parts_dev(rvn_pass1* p, uint32_t n_parts, const rvn_overlap* const* d_parts,
                                    const uint64_t* n_per_part, uint32_t kmax);
int rvn_shard_piles_dev(rvn_engine* e, const uint32_t* lengths /* host */, uint32_t n_reads_total,
                        const rvn_overlap* d_overlaps, const uint32_t* d_overlap_read_off /* n_reads_total + 1 */,
                        uint64_t n, uint32_t kmax, rvn_pass1** out);

/* The two halves of the first step of a round, for the sharded round: reads are mapped independently of
 * each other, so rank g maps the slice [read_first, read_last) of the reads and the ranks all-gather the table.
 *   rvn_polish_map_best  index the targets, map the slice, best overlap per read (racon Polisher::Initialize: longest
 *                        overlap after the error filter); best[i] / best_target[i] describe read read_first + i,
 *                        best_target = index into `targets` or 0xFFFFFFFF when the read is not used.
 *   rvn_polish_set_best  hands the complete table (n_reads = size of the read set) to the NEXT rvn_polish_round[_range]
 *                        call on this engine, which then skips its own mapping; the table is consumed by that call. */
int rvn_polish_map_best(rvn_engine* e, rvn_reads* targets, rvn_reads* reads, uint32_t read_first, uint32_t read_last,
                        double err, rvn_overlap* best, uint32_t* best_target, uint64_t* n_overlaps);
int rvn_polish_set_best(rvn_engine* e, const rvn_overlap* best, const uint32_t* best_target, uint32_t n_reads);

/* Same round restricted to the windows [window_first, window_last) of the global numbering (windows of target 0,
 * then of target 1, ...; ceil(len / w) per target): what one GPU does when a round is sharded by windows
 * (windows are independent).  out_codes receives, per target, the consensus of ITS windows inside the
 * range only (possibly empty), n_windows / n_polished the per-target window counts inside the range; concatenating the
 * per-target pieces of consecutive ranges reproduces rvn_polish_round exactly (raven_amd/sharded.py). */
int rvn_polish_round_range(rvn_engine* e, rvn_reads* targets, rvn_reads* reads, const uint8_t* read_quals,
                           const uint64_t* qual_offsets, double q, double err, uint32_t w, int trim, int match,
                           int mismatch, int gap, uint64_t window_first, uint64_t window_last, uint8_t* out_codes,
                           const uint64_t* out_offsets, uint32_t* out_len, double* ratio, uint32_t* n_windows,
                           uint32_t* n_polished, rvn_polish_stats* stats);

/* ---- N GPUs behind ONE host process (the engine with a device list; 8(e): reads by pile, minimizers by hash
 * class, three exchanges per flush window) --------------------------------------------------------------------------------
 * raven::ConstructGraph / raven::Polish own one ram::MinimizerEngine / one racon::Polisher (RavenLib/src/construct.cc:661-669,
 * polish.cc:43-51), so "all visible GPUs" lives behind one handle: a group = one engine + one worker thread per listed
 * device (a device may be listed more than once: virtual ranks).  The stages are the rvn_shard_* entry points above; the
 * exchanges are pairwise hipMemcpyPeerAsync pulls between the engines' buffers (xGMI between devices) bracketed by
 * in-process barriers — the all-to-all a torch.distributed job performs with RCCL (raven_amd/sharded.py), without leaving
 * the process.  Results are bit-identical to the single-engine calls.
 *   rvn_group_find_overlaps_and_create_piles   = rvn_find_overlaps_and_create_piles over all devices: bounds[n_devices + 1]
 *       receives the read ranges, out[r] a pass handle whose piles / overlap lists are complete for the reads
 *       [bounds[r], bounds[r+1]) (fetch them with rvn_pass1_fetch_* — arrays are indexed by GLOBAL read id — and release
 *       with rvn_pass1_destroy).  A read set beyond 2^32 bases is indexed in several batches as construct.cc:32-37 does
 *       (rvn_group_find_overlaps_and_create_piles_batched takes the batch size: tests force several batches with it).
 *   rvn_group_polish_round                     = rvn_polish_round over all devices (reads mapped by slice, windows by range,
 *       pieces concatenated in rank order); host arrays in, consensus out as in rvn_polish_round. */
typedef struct rvn_group rvn_group;
int rvn_group_create(rvn_group** out, uint32_t k, uint32_t w, uint32_t bandwidth, uint32_t chain, uint32_t matches,
                     uint32_t gap, const int* devices, uint32_t n_devices);
void rvn_group_destroy(rvn_group* g);
uint32_t rvn_group_size(const rvn_group* g);
rvn_engine* rvn_group_engine(rvn_group* g, uint32_t rank);
int rvn_group_find_overlaps_and_create_piles(rvn_group* g, const uint64_t* packed, const uint64_t* word_offsets,
                                             const uint32_t* lengths, uint32_t n_reads, double freq, uint32_t kmax,
                                             int use_minhash, uint64_t flush_bases, uint32_t* bounds, rvn_pass1** out);
int rvn_group_find_overlaps_and_create_piles_batched(rvn_group* g, const uint64_t* packed, const uint64_t* word_offsets,
                                                     const uint32_t* lengths, uint32_t n_reads, double freq, uint32_t kmax,
                                                     int use_minhash, uint64_t index_batch_bases, uint64_t flush_bases,
                                                     uint32_t* bounds, rvn_pass1** out);
int rvn_group_polish_round(rvn_group* g, const uint64_t* t_packed, const uint64_t* t_word_offsets, const uint32_t* t_lengths,
                           uint32_t n_targets, const uint64_t* r_packed, const uint64_t* r_word_offsets,
                           const uint32_t* r_lengths, uint32_t n_reads, double q, double err, uint32_t w, int trim, int match,
                           int mismatch, int gap, uint8_t* out_codes, const uint64_t* out_offsets, uint32_t* out_len,
                           double* ratio);
/* The same round with base qualities (the FASTQ variant of raven::Polish: racon's mean-quality filter against `q`,
 * polish.cc:26-41, and quality-weighted edges in the window graphs): r_quals = Phred+33 bytes, one per
 * 2^qual_block_shift bases of a read (6 = biosoup's block_quality), read i at r_qual_offsets[i] — exactly what
 * rvn_reads_attach_quality takes; every rank attaches them to its copy of the read set.  r_quals == NULL is
 * rvn_group_polish_round. */
int rvn_group_polish_round_q(rvn_group* g, const uint64_t* t_packed, const uint64_t* t_word_offsets, const uint32_t* t_lengths,
                             uint32_t n_targets, const uint64_t* r_packed, const uint64_t* r_word_offsets,
                             const uint32_t* r_lengths, uint32_t n_reads, const uint8_t* r_quals, const uint64_t* r_qual_offsets,
                             int qual_block_shift, double q, double err, uint32_t w, int trim, int match, int mismatch, int gap,
                             uint8_t* out_codes, const uint64_t* out_offsets, uint32_t* out_len, double* ratio);
/* Which ranks reach each other's memory directly: direct[i * n + j] = 1 when rank i pulls from rank j without staging (the
 * same device, or hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess succeeded: xGMI on an MI355X node); 0 = the copy goes
 * through host memory (still correct, PCIe speed).  direct may be NULL.  Returns 1 when every pair is direct, 0 when
 * some pair is not, RVN_EINVAL for a NULL group. */
int rvn_group_peer_access(const rvn_group* g, uint8_t* direct);

/* raven::OverlapUpdate (RavenLib/src/overlap_utils.cc:14-85) followed by raven::GetOverlapType (:87-121) on a list of
 * overlaps, on the HOST (the library's __host__ build of the rules its kernels run: raven_amd/csrc/overlap_rules.h) — for
 * a caller that holds overlaps and pile regions in host arrays between two device stages, as ResolveContainedReads does
 * (construct.cc:218-254).  pile_begin / pile_end in bases, pile_invalid != 0 = Pile::is_invalid().  ok[i] = OverlapUpdate's
 * result (the overlap is updated in place when 1); type[i] = GetOverlapType of the updated overlap (0 internal, 1 lhs
 * contained, 2 rhs contained, 3 / 4 dovetails), 0xFFFFFFFF when ok[i] == 0. */
int rvn_overlap_update_and_type(rvn_overlap* overlaps, uint64_t n, const uint32_t* pile_begin, const uint32_t* pile_end,
                                const uint8_t* pile_invalid, uint32_t n_piles, uint8_t* ok, uint32_t* type);

/* Tuning a deployment may set; -1 restores the built-in default of any option (so does 0, except for poa_rows_min_windows).
 * No option changes an overlap list, a pile or a layer table; poa_rows_min_windows chooses which window-consensus kernel
 * makes the first attempt, and the two agree on 19 998 of 20 000 C4-like windows, not on every one: a
 * consensus is reproducible for a fixed value of that option, not across values.  The product library reads
 * NO environment variable that alters what a call computes or how it is scheduled (the only ones it reads at all:
 * RVN_EDLIB_DEVICE of the edlibAlign drop-in, RVN_DEVICES of include/raven_hip/multi_gpu.hpp — which device); debugging
 * switches exist only in libraven_hip_test.so (built with -DRVN_DEBUG_KNOBS).  Options:
 *   nw_budget_mb       alignment-path stage: HBM for the stored band words (default: a quarter of the free memory, <= 64 GB)
 *   nw_group_walk      alignment-path stage: which walk a launch takes — 1 a lane per alignment, 2 a group of sixteen lanes per
 *                      alignment, 3 a lane per alignment with half-size strips, otherwise (default) the group for launches of at
 *                      most 8 192 alignments and the half-size strips beyond 65 536; same records either way
 *   index_direct_min_keys  index: distinct values from which all 4^k possible values are addressed directly (an 8-GB table at
 *                      k = 15, one cache line per probe; default 8 388 608; 1 = every index with k <= 15); same matches either way
 *   poa_rows_min_windows  window-consensus stage: smallest batch that starts with the rows-on-lanes kernel (default 8 192;
 *                      0 means every batch, -1 the default)
 *   io_threads, io_slab_mb, io_ring, io_zlib   rvn_reads_load: inflate threads, page-locked slab size, slabs in flight,
 *                      != 0: zlib instead of this library's own inflate on a single gzip member
 *   arena_mb, arena_margin_mb, no_arena, release_always   the device arena behind the scratch buffers
 *   polish_sketch_cache_mb  polishing rounds: HBM for the reads' sketch kept from one round to the next (the reads do not change
 *                      between rounds; default an eighth of the device, 0 = recompute every round)
 *   polish_join        != 0: a round maps by sorting the reads' minimizers with the targets' and streaming the runs instead of
 *                      probing the targets' index — same overlaps, slower at the metric's size
 * previous (may be NULL) receives the value the option had (its default's value when it was at the default).  RVN_EINVAL:
 * unknown name, a value below -1, io_ring == 1 (a ring needs two slabs). */
int rvn_engine_set_option(rvn_engine* e, const char* name, int64_t value, int64_t* previous);

/* Kept for source compatibility: a round no longer has a host cutting stage to overlap with the POA, all windows
 * of a call are one device batch.  Stores the value, returns the previous one; results never depended on it. */
uint64_t rvn_polish_set_chunk_windows(rvn_engine* e, uint64_t windows);

/* The window layers of the last rvn_polish_round / _range call, as racon would have added them (steps 1-4 of its
 * round: mapping, best overlap, alignment path, breakpoints, layer rules — all integer work, compared bit-exactly with
 * the CPU restatement by the tests): 7 uint32 per layer {global window, read index, first base in the oriented read,
 * bases, begin, end, reverse-complemented}, windows in order, layers in racon's order, layers dropped by the
 * mean-quality filter left out.  out == NULL or cap too small: only *n is set. */
int rvn_polish_fetch_layers(rvn_engine* e, uint32_t* out, uint64_t cap, uint64_t* n);

/* reads used per target (their best overlap passed the error filter) in the last rvn_polish_round call: the RC:i:
 * tag racon writes next to XC:f: */
int rvn_polish_target_reads(const rvn_engine* e, uint32_t* counts, uint32_t n_targets);

/* shader-clock cycles summed over all windows of the last rvn_poa_consensus_batch call, per phase:
 * {subgraph, NW matrix, traceback, AddAlignment, order rebuild, consensus} */
void rvn_poa_phase_cycles(const rvn_engine* e, uint64_t out[6]);

/* DP work of the banded window kernel since the last rvn_engine_reset_stats: {cells of the full-matrix equivalent =
 * graph rows x layer length over every layer alignment (what spoa's NW computes; windows repeated with a wider band
 * count again), cells inside the computed bands, POA batches}.  bench.py prices the kernel's roofline with it. */
void rvn_poa_work(const rvn_engine* e, uint64_t out[3]);

/* Which window kernel rvn_poa_consensus_batch / rvn_polish_round use: 0 (default) = the rows-on-lanes kernel with a
 * 32-column band (poa4.hip); windows whose alignment touches the band edge, or whose graph is beyond that kernel's
 * limits, are repeated by the one-row-per-iteration kernel (poa2.hip) with 64, then 128 and 256 columns, and what is
 * left (or beyond a limit) by the full-matrix kernel; 1 = full-matrix kernel only; 2 / 3 / 4 = 64- / 128- / 256-column
 * band only (flagged windows come back with status 8); 9 = poa4.hip only.  Returns the previous mode. */
int rvn_poa_set_mode(rvn_engine* e, int mode);
/* mode 0: windows of the last batch repeated with the 128-column band (wide) / that needed more than that (fallback:
 * 256 columns or the full matrix) */
uint32_t rvn_poa_fallback_windows(const rvn_engine* e);
uint32_t rvn_poa_wide_windows(const rvn_engine* e);
/* mode 0: windows of the last batch that the 32-column first attempt (poa4.hip) handed on to the 64-column kernel */
uint32_t rvn_poa_narrow_windows(const rvn_engine* e);

/* ---- introspection used by the parity tests and bench.py ------------------------------------- */
/* sketch of reads [first,last) == ram Minimize(sequence, minhash) per read; values widened to u64 */
int rvn_engine_sketch(rvn_engine* e, const rvn_reads* r, uint32_t first, uint32_t last, int minhash,
                      uint64_t* count);
int rvn_engine_sketch_fetch(rvn_engine* e, uint64_t* values, uint64_t* origins, uint32_t* read_offsets);
/* sorted index content: (value, origin) pairs in index order; count = minimizers in the index */
int rvn_engine_index_size(const rvn_engine* e, uint64_t* n_minimizers, uint64_t* n_keys);
int rvn_engine_index_fetch(rvn_engine* e, uint64_t* values, uint64_t* origins);
/* counters since the last reset: {index_bases, index_minimizers, index_keys, query_bases,
 * query_minimizers, matches, overlaps(Map outputs), intervals} */
int rvn_engine_counters(const rvn_engine* e, uint64_t out[8]);
/* accumulated device milliseconds per stage (HIP events on the engine's stream) and launch counts */
int rvn_engine_num_stages(void);
const char* rvn_engine_stage_name(int stage);
int rvn_engine_stage_ms(const rvn_engine* e, double* ms, uint64_t* launches, int n);
void rvn_engine_reset_stats(rvn_engine* e);
void rvn_engine_set_timing(rvn_engine* e, int enabled);
/* per-kernel-site device time: HIP events recorded on the engine's stream around every launch of the
 * site (no host synchronisation while recording); rvn_engine_kernel_ms synchronises and accumulates. */
void rvn_engine_set_kernel_timing(rvn_engine* e, int enabled);
int rvn_engine_num_kernel_sites(void);
const char* rvn_engine_kernel_site_name(int site);
int rvn_engine_kernel_ms(rvn_engine* e, double* ms, uint64_t* launches, int n);

/* The host-side test hooks (rvn_test_*, rvn_poa_banded_emulate) are NOT part of this library: include/raven_hip_test.h,
 * libraven_hip_test.so. */

#ifdef __cplusplus
}
#endif
#endif /* RAVEN_HIP_H_ */
