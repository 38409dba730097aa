// io.hip — the input path of the overlap phase: FASTA / FASTQ (optionally gzip'ed) -> packed reads resident in HBM
// (what raven gets from raven::CreateParser + Parse(-1) into biosoup::NucleicAcid objects: RavenLib/src/io.cc:7-41,
// RavenExe/src/main.cc:258-299; bioparser / biosoup are not in the reference tree, their behaviour is restated from the
// call sites and SURVEY.md App. A.4).
//
//   inflate pool (io_text.h)  the file is mapped and, if gzip'ed, cut into its members (BGZF blocks, concatenated
//                             members); N threads inflate them straight into their place of the TEXT, which lives in a
//                             ring of page-locked slabs.  A single-member archive is one stream: one thread, front to back.
//   caller thread             per completed slab: one memchr pass finds the records (RecordScanner: no copy on the host
//                             unless a sequence is wrapped over several lines), the slab goes to HBM as it is (async
//                             H2D), and every ~128 MB of text the records are cut out ON THE DEVICE: ASCII -> 2-bit
//                             codes through biosoup's coder table (IUPAC folded, anything else rejected), 32 bases per
//                             word, every read on a word boundary, appended to the growing packed array; FASTQ: mean
//                             quality of every 64-base block (biosoup's block_quality), attached to the read set for the
//                             polishing rounds.
// Inflating slab k+R, scanning slab k+1 and copying slab k overlap; the bases never exist on the host in packed form.
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"
#include "io_text.h"

namespace rvn {

namespace {

__constant__ u8 c_coder[256];

void coder_table(u8* t) {  // biosoup::NucleicAcid's coder: 255 = not a nucleotide
  std::memset(t, 255, 256);
  const char* groups[4] = {"AaDdNnRrWw-", "CcBbMmSs", "GgKkVv", "TtUuHhYy"};
  for (int c = 0; c < 4; ++c)
    for (const char* p = groups[c]; *p; ++p) t[static_cast<unsigned char>(*p)] = static_cast<u8>(c);
}

// one thread per output word of the chunk: chars -> codes -> 32 bases per word
// read i = chars[field_off[i] .. field_off[i] + len[i]) of the text
__global__ void pack_ascii_kernel(const u8* __restrict__ chars, const u64* __restrict__ field_off,
                                  const u32* __restrict__ len, const u64* __restrict__ word_off, u32 n_reads, u64 n_words,
                                  u64* __restrict__ packed, u32* __restrict__ bad) {
  const u64 wi = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (wi >= n_words) return;
  u32 lo = 0, hi = n_reads;  // read of this word: last i with word_off[i] <= wi (word_off relative to the chunk)
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (word_off[mid] <= wi) lo = mid;
    else hi = mid;
  }
  const u64 first = field_off[lo] + (wi - word_off[lo]) * 32;
  const u64 end = field_off[lo] + len[lo];
  u64 w = 0;
  for (u32 x = 0; x < 32 && first + x < end; ++x) {
    const u8 code = c_coder[chars[first + x]];
    if (code > 3) {
      atomicAdd(bad, 1u);
    } else {
      w |= static_cast<u64>(code) << (2 * x);
    }
  }
  packed[wi] = w;
}

// biosoup block_quality: integer mean of (q - '!') over every 64-base block; stored + 33 (what the polishing rounds read)
__global__ void block_quality_kernel(const u8* __restrict__ quals, const u64* __restrict__ field_off,
                                     const u32* __restrict__ len, const u64* __restrict__ blk_off, u32 n_reads,
                                     u64 n_blocks, u8* __restrict__ out) {
  const u64 bi = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (bi >= n_blocks) return;
  u32 lo = 0, hi = n_reads;
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (blk_off[mid] <= bi) lo = mid;
    else hi = mid;
  }
  const u64 first = field_off[lo] + (bi - blk_off[lo]) * 64;
  const u64 end = field_off[lo] + len[lo];
  u32 sum = 0, cnt = 0;
  for (u32 x = 0; x < 64 && first + x < end; ++x) {
    sum += static_cast<u32>(quals[first + x]) - 33u;
    ++cnt;
  }
  out[bi] = static_cast<u8>((cnt ? sum / cnt : 0u) + 33u);
}

// grow-preserving device append
void ensure_capacity(DevBuf& buf, u64 used_bytes, u64 need_bytes, hipStream_t s) {
  if (need_bytes <= buf.cap) return;
  DevBuf bigger;
  bigger.reserve(std::max<u64>(need_bytes * 2, 1 << 20));
  if (used_bytes) RVN_HIP(hipMemcpyAsync(bigger.ptr, buf.ptr, used_bytes, hipMemcpyDeviceToDevice, s));
  RVN_HIP(rvn_stream_sync(s));
  std::swap(buf.ptr, bigger.ptr);
  std::swap(buf.cap, bigger.cap);
}

bool has_suffix(const std::string& s, const char* suf) {
  const size_t n = std::strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

}  // namespace


namespace {

// One pass over the file.  streaming: one inflate thread front to back (second attempt after a wrong member cut).
void load_once(Engine& e, const std::string& path, bool fastq, bool streaming, bool& speculated, ReadsDev& R, std::vector<std::string>& names,
               LoadStats& st) {
  auto secs = [](std::chrono::steady_clock::time_point a) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
  };
  hipStream_t s = e.stream;
  io::SourceOptions opt;
  opt.force_streaming = streaming;
  opt.threads = static_cast<u32>(e.opt.io_threads);
  if (e.opt.io_slab_mb > 0) opt.slab_bytes = static_cast<u64>(e.opt.io_slab_mb) << 20;
  if (e.opt.io_ring > 1) opt.ring = static_cast<u32>(e.opt.io_ring);
  opt.zlib_only = e.opt.io_zlib != 0;
  opt.debug = knob("RVN_IO_DEBUG") != nullptr;
  // the slabs are page-locked once per engine and handed out again by later loads (pinning runs at 1-2 GB/s)
  opt.alloc = [&e](size_t n) -> void* {
    for (auto& slot : e.io_pin)
      if (!slot.second && slot.first->cap >= n) {
        slot.second = true;
        return slot.first->ptr;
      }
    e.io_pin.emplace_back(std::unique_ptr<PinBuf>(new PinBuf()), true);
    return e.io_pin.back().first->get<u8>(n);
  };
  opt.release = [&e](void* p) {
    for (auto& slot : e.io_pin)
      if (slot.first->ptr == p) slot.second = false;
  };
  speculated = false;
  io::TextSource src(path, opt);
  // did this attempt speculate?  (member cuts, or this repository's own decoder on a single stream: only then can an
  // error be the speculation's fault and a second attempt tell anything new)
  speculated = src.gzip() && (!src.streaming() || src.fast_stream());
  io::RecordScanner scanner(fastq);

  names.clear();
  R.h_len.clear();
  R.total_bases = 0;
  R.h_word_off.assign(1, 0);
  std::vector<u64> h_qoff(1, 0);
  u64 words_used = 0, qblocks_used = 0;
  DevBuf d_foff, d_qfoff, d_len, d_woff, d_qoff;
  u32* d_bad = e.tmp_f.get<u32>(4);
  RVN_HIP(hipMemsetAsync(d_bad, 0, 4, s));

  // the kept text in HBM: d_text[cur][0] is offset text_base of the kept text
  int cur = 0;
  u64 text_base = 0;
  constexpr u64 kShipBytes = 128ULL << 20;
  std::vector<io::TextRecord> recs;
  std::vector<std::string> rec_names;
  std::vector<u64> foff, qfoff, woff, qoff;
  std::vector<u32> lens;
  double scan_busy = 0, dev_busy = 0;

  // records completed so far -> packed words (+ block qualities), then only the text of the record in progress is kept
  auto ship = [&](bool last) {
    const u32 nr = static_cast<u32>(recs.size());
    const u64 text_end = scanner.text_end();
    if (nr) {
      foff.resize(nr);
      qfoff.resize(nr);
      lens.resize(nr);
      woff.assign(nr + 1, 0);
      qoff.assign(nr + 1, 0);
      for (u32 i = 0; i < nr; ++i) {
        foff[i] = recs[i].seq_off - text_base;
        qfoff[i] = fastq ? recs[i].qual_off - text_base : 0;
        lens[i] = static_cast<u32>(recs[i].len);
        woff[i + 1] = woff[i] + (recs[i].len + 31) / 32;
        qoff[i + 1] = qoff[i] + (recs[i].len + 63) / 64;
      }
      const u64 nw = woff[nr], nq = qoff[nr];
      const u8* text = e.io_text[cur].as<u8>();
      u64* dfo = d_foff.get<u64>(nr);
      u32* dln = d_len.get<u32>(nr);
      u64* dwo = d_woff.get<u64>(nr + 1);
      RVN_HIP(hipMemcpyAsync(dfo, foff.data(), nr * 8ULL, hipMemcpyHostToDevice, s));
      RVN_HIP(hipMemcpyAsync(dln, lens.data(), nr * 4ULL, hipMemcpyHostToDevice, s));
      RVN_HIP(hipMemcpyAsync(dwo, woff.data(), (nr + 1) * 8ULL, hipMemcpyHostToDevice, s));
      ensure_capacity(R.packed, words_used * 8, (words_used + nw + 2) * 8, s);
      if (nw) {
        pack_ascii_kernel<<<div_up(nw, 256), 256, 0, s>>>(text, dfo, dln, dwo, nr, nw, R.packed.as<u64>() + words_used, d_bad);
        RVN_LAUNCH_CHECK();
      }
      if (fastq) {
        u64* dqf = d_qfoff.get<u64>(nr);
        u64* dqo = d_qoff.get<u64>(nr + 1);
        RVN_HIP(hipMemcpyAsync(dqf, qfoff.data(), nr * 8ULL, hipMemcpyHostToDevice, s));
        RVN_HIP(hipMemcpyAsync(dqo, qoff.data(), (nr + 1) * 8ULL, hipMemcpyHostToDevice, s));
        ensure_capacity(R.quals, qblocks_used, qblocks_used + nq + 16, s);
        if (nq) {
          block_quality_kernel<<<div_up(nq, 256), 256, 0, s>>>(text, dqf, dln, dqo, nr, nq, R.quals.as<u8>() + qblocks_used);
          RVN_LAUNCH_CHECK();
        }
      }
      for (u32 i = 0; i < nr; ++i) {
        R.h_len.push_back(lens[i]);
        R.h_word_off.push_back(words_used + woff[i + 1]);
        h_qoff.push_back(qblocks_used + qoff[i + 1]);
        R.total_bases += lens[i];
        names.emplace_back(std::move(rec_names[i]));
      }
      words_used += nw;
      qblocks_used += nq;
      recs.clear();
      rec_names.clear();
    }
    if (!last) {  // the record in progress moves to the front of the other text buffer
      const u64 from = std::max(text_base, std::min(scanner.retain_from(), text_end));
      const u64 tail = text_end - from;
      e.io_text[cur ^ 1].reserve(std::max<u64>(tail + (64ULL << 20), kShipBytes + (64ULL << 20)));
      if (tail)
        RVN_HIP(hipMemcpyAsync(e.io_text[cur ^ 1].ptr, e.io_text[cur].as<u8>() + (from - text_base), tail,
                               hipMemcpyDeviceToDevice, s));
      text_base = from;
      cur ^= 1;
    }
    RVN_HIP(rvn_stream_sync(s));  // the offset arrays above are free again
  };

  hipEvent_t ev[2];
  RVN_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
  RVN_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  struct EvGuard {
    hipEvent_t* ev;
    hipStream_t s;
    ~EvGuard() {
      (void)hipStreamSynchronize(s);  // no copy may still read a slab when the source hands them back
      (void)hipEventDestroy(ev[0]);
      (void)hipEventDestroy(ev[1]);
    }
  } guard{ev, s};
  e.io_text[0].reserve(kShipBytes + (64ULL << 20));

  u8* slab = nullptr;
  u64 n = 0, k = 0;
  while (src.next(&slab, &n)) {
    const auto t0 = std::chrono::steady_clock::now();
    const u8* run = nullptr;
    u64 run_len = 0, run_base = 0;
    scanner.scan(slab, n, &run, &run_len, &run_base, recs, rec_names);
    scan_busy += secs(t0);
    const auto t1 = std::chrono::steady_clock::now();
    if (run_base < text_base) throw std::logic_error("[raven_hip] input path: run before the retained text");
    ensure_capacity(e.io_text[cur], run_base - text_base, run_base - text_base + run_len + 64, s);
    if (run_len)
      RVN_HIP(hipMemcpyAsync(e.io_text[cur].as<u8>() + (run_base - text_base), run, run_len, hipMemcpyHostToDevice, s));
    RVN_HIP(hipEventRecord(ev[k & 1], s));
    if (k > 0) {  // slab k-1 has been copied: the pool may overwrite it (slab k is still in flight)
      RVN_HIP(hipEventSynchronize(ev[(k - 1) & 1]));
      src.release();
    }
    ++k;
    if (scanner.text_end() - text_base >= kShipBytes && !recs.empty()) ship(false);
    dev_busy += secs(t1);
  }
  {
    u8 extra = 0;
    const u64 at = scanner.text_end();
    if (scanner.finish(recs, rec_names, &extra)) {
      ensure_capacity(e.io_text[cur], at - text_base, at - text_base + 64, s);
      RVN_HIP(hipMemcpy(e.io_text[cur].as<u8>() + (at - text_base), &extra, 1, hipMemcpyHostToDevice));
    }
    const auto t1 = std::chrono::steady_clock::now();
    ship(true);
    dev_busy += secs(t1);
  }
  if (read_back(e, d_bad, 4) != 0)
    throw std::invalid_argument("[biosoup::NucleicAcid::NucleicAcid] error: not a nucleotide");
  // finish the read set exactly as rvn_reads_upload does
  const u32 nseq = static_cast<u32>(R.h_len.size());
  if (R.h_len.size() >= (1ULL << 30)) throw std::invalid_argument("[raven_hip] more than 2^30 sequences");
  R.n = nseq;
  R.n_words = words_used;
  R.h_id.resize(nseq);
  for (u32 i = 0; i < nseq; ++i) R.h_id[i] = i;
  R.ids_are_indices = true;
  ensure_capacity(R.packed, words_used * 8, (words_used + 2) * 8, s);
  RVN_HIP(hipMemsetAsync(R.packed.as<u64>() + words_used, 0, 16, s));
  u64* d_wo = R.word_off.get<u64>(static_cast<size_t>(nseq) + 1);
  u32* d_ln = R.len.get<u32>(static_cast<size_t>(nseq) + 1);
  u32* d_id = R.id.get<u32>(static_cast<size_t>(nseq) + 1);
  RVN_HIP(hipMemcpyAsync(d_wo, R.h_word_off.data(), (static_cast<size_t>(nseq) + 1) * 8, hipMemcpyHostToDevice, s));
  if (nseq) {
    RVN_HIP(hipMemcpyAsync(d_ln, R.h_len.data(), static_cast<size_t>(nseq) * 4, hipMemcpyHostToDevice, s));
    RVN_HIP(hipMemcpyAsync(d_id, R.h_id.data(), static_cast<size_t>(nseq) * 4, hipMemcpyHostToDevice, s));
  }
  if (fastq) {
    u64* d_qo = R.qual_off.get<u64>(static_cast<size_t>(nseq) + 1);
    RVN_HIP(hipMemcpyAsync(d_qo, h_qoff.data(), (static_cast<size_t>(nseq) + 1) * 8, hipMemcpyHostToDevice, s));
    R.h_qual_off = h_qoff;
    R.qual_shift = 6;
  }
  RVN_HIP(rvn_stream_sync(s));
  reads_build_tiles(e, R);
  st.n_sequences = nseq;
  st.n_bases = R.total_bases;
  st.has_quality = fastq ? 1 : 0;
  st.parse_s = scan_busy;
  st.device_s = dev_busy;
  st.inflate_threads = src.threads();
  st.members = src.members();
  st.streaming = src.streaming() ? 1 : 0;
}

}  // namespace

// Loads a sequence file into `R` (ids = 0 .. n-1 in file order, like biosoup's num_objects counter starting at 0).
// names: the sequences' names (first word of the header).  Throws std::invalid_argument for an unsupported extension, an
// unreadable file, a malformed record, a damaged archive or a character that is not a nucleotide (biosoup's own error).
void reads_load(Engine& e, const std::string& path, ReadsDev& R, std::vector<std::string>& names, LoadStats& st) {
  st = LoadStats();
  const auto t_all = std::chrono::steady_clock::now();
  bool fastq;
  if (has_suffix(path, ".fasta") || has_suffix(path, ".fa") || has_suffix(path, ".fasta.gz") || has_suffix(path, ".fa.gz")) fastq = false;
  else if (has_suffix(path, ".fastq") || has_suffix(path, ".fq") || has_suffix(path, ".fastq.gz") || has_suffix(path, ".fq.gz")) fastq = true;
  else
    throw std::invalid_argument("[raven::CreateParser] error: file " + path +
                                " has unsupported format extension (valid extensions: .fasta, .fasta.gz, .fa, .fa.gz, "
                                ".fastq, .fastq.gz, .fq, .fq.gz)");
  {
    u8 table[256];
    coder_table(table);
    RVN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_coder), table, 256));
  }
  // The first attempt speculates (member cuts, this repository's own decoder for a single stream).  Whatever goes wrong
  // there — a cut that was no member boundary, a member that does not verify, but also a record or a character the
  // scanner / the device refuse in text that speculation produced — is settled by the second attempt: zlib, front to back.
  // An error is only ever reported from that one (or from a first attempt that found nothing to doubt).
  bool again = false, speculated = false;
  try {
    load_once(e, path, fastq, false, speculated, R, names, st);
  } catch (const io::SpeculationFailed&) {
    again = true;
  } catch (const std::invalid_argument&) {
    // a plain file, a file that cannot be opened, or a gzip stream zlib itself was reading: nothing was speculated, the
    // error is the input's and a second pass would only reproduce it
    if (!speculated) throw;
    again = true;
  }
  if (again) {
    bool unused = false;
    try {
      load_once(e, path, fastq, true, unused, R, names, st);
    } catch (const io::SpeculationFailed&) {
      throw std::invalid_argument("[bioparser] error: corrupt or truncated file");
    }
    st.restarted = 1;
  }
  st.total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
}

}  // namespace rvn
