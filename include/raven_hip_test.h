/* raven_hip_test.h — TEST INFRASTRUCTURE, not part of the product: the host-side hooks of libraven_hip_test.so.
 *
 * libraven_hip_test.so = the objects of libraven_hip.so with engine / poa / poa4 / nwpath compiled again under
 * -DRVN_TEST_HOOKS, plus the host wavefront emulator (simt_emu.hip).  It exports everything raven_hip.h declares and,
 * in addition, the entry points below, which step the __host__ __device__ building blocks of the kernels on the CPU
 * so that the CPU suite (pytest -m "not gpu") can compare them with the oracle without a GPU.  Nothing on the product
 * path calls them and libraven_hip.so does not export them (tests/test_abi.py checks both). */
#ifndef RAVEN_HIP_TEST_H_
#define RAVEN_HIP_TEST_H_

#include "raven_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The rows-on-lanes banded kernel (poa4.hip) stepped through on the HOST by a 64-fibre
 * wavefront emulator — the same phase functions, no GPU and no engine needed.  Arguments as rvn_poa_consensus_batch +
 * variant (ignored: one kernel); first attempt only (status 8: the window needs a wider band).  The CPU suite
 * compares it with the oracle (tests/test_poa4_emulation.py); nothing on the product path calls it. */
int rvn_poa_banded_emulate(const uint8_t* codes, const uint8_t* quals, const uint64_t* layer_offsets,
                           const uint32_t* begins, const uint32_t* ends, const uint32_t* has_qual,
                           const uint32_t* window_offsets, uint32_t n_windows, int match, int mismatch, int gap,
                           int trim, uint8_t* consensus, const uint64_t* consensus_offsets, uint32_t* consensus_len,
                           uint32_t* status, int variant);

/* host-side test hooks for the __host__ __device__ building blocks (no GPU needed) */
uint64_t rvn_test_hash(uint64_t key, uint32_t k, int use32);
int rvn_test_canonical(const uint64_t* words, uint32_t pos, uint32_t k, int use32, uint64_t* value, uint32_t* strand);
int rvn_test_low_complexity(const uint8_t* codes, uint32_t k);
/* the alignment-path stage of a polishing round (nwpath.h) stepped on the CPU: the forward sweep's lane code driven
 * for 64 emulated lanes + the traceback, i.e. exactly what the kernels execute.  Rows = target span
 * [t_begin, t_begin + n) of a packed target, columns = span [q_begin, q_begin + m) of the read in the target's
 * orientation (rc: the read is reverse-complemented).  k = first band threshold (doubled until exact), force_r = 0
 * or the blocks per lane (1 / 2 / 4 / 8).  recs: one 32-byte record per window of w target bases touched by the span
 * {first_t, first_q, last_t, last_q, u16 grid[8]}; distance = exact edit distance; band = {k, lanes, R} used.
 * Bits 8-15 of rc: 0 = the walk of one lane per alignment (whole strips), 1 = the same with strips of sixteen kept columns (what
 * the kernel runs where the strips live in LDS); 4 / 16 / 64 = the group walk with that many lanes per alignment
 * (nwtrace.h: NwGroupWalk, its phases stepped lane by lane) — band then has a fourth entry, the batches the walk took.
 * Returns 0, 1 if the walk did not end at cost 0, < 0 on invalid arguments. */
int rvn_test_nw_breakpoints(const uint64_t* t_words, uint32_t t_len, const uint64_t* r_words, uint32_t r_len,
                            uint32_t t_begin, uint32_t n, uint32_t q_begin, uint32_t m, int rc, uint32_t w, uint32_t k,
                            int force_r, uint32_t* recs, uint32_t* distance, uint32_t* band);
/* Pile::FindChimericRegions (slopes.h, the __host__ __device__ code the kernel runs) on one coverage array: out = (begin,
 * end) cell pairs; returns their number, -5 if a capacity was exceeded */
int64_t rvn_test_find_chimeric_regions(const uint16_t* data, uint32_t size, uint32_t* out, uint64_t cap_pairs);
/* OverlapUpdate + GetOverlapType (overlap_rules.h, the __host__ __device__ code the kernels run) on a list: ok[i] =
 * OverlapUpdate result (the overlap is updated in place when ok), type[i] = GetOverlapType of the updated overlap */
int rvn_test_overlap_update_and_type(rvn_overlap* overlaps, uint64_t n, const uint32_t* pile_begin, const uint32_t* pile_end,
                                     const uint8_t* pile_invalid, uint32_t n_piles, uint8_t* ok, uint32_t* type);
/* The host half of rvn_reads_load (io_text.h: gzip member cut, inflate pool, FASTA / FASTQ record scanner) without a
 * device: fastq 0 / 1; threads 0 = default; force_streaming: one inflate thread front to back; slab_bytes 0 = default.
 * Outputs malloc'ed (free with rvn_free): all bases back to back, all qualities (FASTQ), lengths[n_records], names
 * separated by '\n'; info[8] = {gzip, streaming, members found, pool threads, 1 if a wrong cut made it start over,
 * microseconds of the inflate + scan loop, of which inside the scanner, 1 if the single stream went through
 * inflate_fast.h}. */
int rvn_test_parse_file(const char* path, int fastq, uint32_t threads, int force_streaming, uint64_t slab_bytes,
                        uint8_t** bases, uint8_t** quals, uint32_t** lengths, uint32_t* n_records, char** names,
                        uint32_t* info);
/* inflate_fast.h (the single-stream deflate decoder of the input path) on ONE gzip member: dst gets the text, out[4] =
 * {bytes produced, bytes of the member consumed incl. its trailer, CRC-32 and ISIZE found in the trailer}; chunk > 0:
 * through a drained buffer of that many bytes (as the input path runs it).  RVN_EINVAL + message for an invalid stream. */
int rvn_test_inflate_fast(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t chunk, uint64_t* out);
/* freelist.h — the offset bookkeeping of the device arena behind the engine's grow-only buffers — driven by a list of
 * operations: ops[i] > 0 allocates that many bytes (out[i] = offset, -1 if no hole holds it), ops[i] <= 0 gives back the
 * block allocated by operation -ops[i] (out[i] = 1, or 0 if it was not in use); state[3] = {bytes free, largest hole,
 * blocks in use} afterwards. */
int rvn_test_freelist(uint64_t size, uint64_t grain, const int64_t* ops, uint32_t n_ops, int64_t* out, uint64_t* state);
void rvn_test_std_sort_lendesc(uint64_t* data, uint64_t n);
void rvn_test_heap_sort_lendesc(uint64_t* data, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* RAVEN_HIP_TEST_H_ */
