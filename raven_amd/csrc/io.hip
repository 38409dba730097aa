// io.hip — the input path of the overlap phase: FASTA / FASTQ (optionally gzip'ed) -> packed reads resident in HBM
// (what raven gets from raven::CreateParser + Parse(-1) into biosoup::NucleicAcid objects: RavenLib/src/io.cc:7-41,
// RavenExe/src/main.cc:258-299; bioparser / biosoup are not in the reference tree, their behaviour is restated from the
// call sites and SURVEY.md App. A.4).
//
//   host thread (producer)   zlib inflate + line parser -> raw base characters (and quality characters) of whole reads
//                            into one of two pinned staging buffers
//   caller thread (consumer) async H2D of a staging buffer, then on the device: ASCII -> 2-bit codes through biosoup's
//                            coder table (IUPAC folded, anything else rejected), 32 bases per word, every read on a word
//                            boundary, appended to the growing packed array; FASTQ: mean quality of every 64-base block
//                            (biosoup's block_quality), attached to the read set for the polishing rounds
// Parsing of chunk i+1 overlaps copy + packing of chunk i; the bases never exist on the host in packed form.
#include <zlib.h>

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
__global__ void pack_ascii_kernel(const u8* __restrict__ chars, const u64* __restrict__ base_off,
                                  const u64* __restrict__ word_off, u32 n_reads, u64 n_words, u64* __restrict__ packed,
                                  u32* __restrict__ bad) {
  const u64 wi = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (wi >= n_words) return;
  u32 lo = 0, hi = n_reads;  // read of this word: last i with word_off[i] <= wi (word_off relative to the chunk)
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (word_off[mid] <= wi) lo = mid;
    else hi = mid;
  }
  const u64 first = base_off[lo] + (wi - word_off[lo]) * 32;
  const u64 end = base_off[lo + 1];
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
__global__ void block_quality_kernel(const u8* __restrict__ quals, const u64* __restrict__ base_off,
                                     const u64* __restrict__ blk_off, u32 n_reads, u64 n_blocks, u8* __restrict__ out) {
  const u64 bi = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (bi >= n_blocks) return;
  u32 lo = 0, hi = n_reads;
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (blk_off[mid] <= bi) lo = mid;
    else hi = mid;
  }
  const u64 first = base_off[lo] + (bi - blk_off[lo]) * 64;
  const u64 end = base_off[lo + 1];
  u32 sum = 0, cnt = 0;
  for (u32 x = 0; x < 64 && first + x < end; ++x) {
    sum += static_cast<u32>(quals[first + x]) - 33u;
    ++cnt;
  }
  out[bi] = static_cast<u8>((cnt ? sum / cnt : 0u) + 33u);
}

struct Chunk {
  PinBuf chars, quals;  // pinned staging
  u64 n_chars = 0;
  std::vector<u32> lengths;
  std::vector<std::string> names;
  bool last = false;
  std::string error;
};

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

// Loads a sequence file into `R` (ids = 0 .. n-1 in file order, like biosoup's num_objects counter starting at 0).
// names: the sequences' names (first word of the header).  Throws std::invalid_argument for an unsupported extension, an
// unreadable file, a malformed record or a character that is not a nucleotide (biosoup's own error).
void reads_load(Engine& e, const std::string& path, ReadsDev& R, std::vector<std::string>& names, LoadStats& st) {
  st = LoadStats();
  const auto t_all = std::chrono::steady_clock::now();
  auto secs = [](std::chrono::steady_clock::time_point a) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
  };
  bool fastq;
  if (has_suffix(path, ".fasta") || has_suffix(path, ".fa") || has_suffix(path, ".fasta.gz") || has_suffix(path, ".fa.gz")) fastq = false;
  else if (has_suffix(path, ".fastq") || has_suffix(path, ".fq") || has_suffix(path, ".fastq.gz") || has_suffix(path, ".fq.gz")) fastq = true;
  else
    throw std::invalid_argument("[raven::CreateParser] error: file " + path +
                                " has unsupported format extension (valid extensions: .fasta, .fasta.gz, .fa, .fa.gz, "
                                ".fastq, .fastq.gz, .fq, .fq.gz)");
  gzFile gz = gzopen(path.c_str(), "rb");
  if (!gz) throw std::invalid_argument("[bioparser::Parser::Create] error: unable to open file " + path);
  gzbuffer(gz, 1 << 20);

  hipStream_t s = e.stream;
  {
    u8 table[256];
    coder_table(table);
    RVN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_coder), table, 256));
  }
  constexpr u64 kChunkChars = 64ULL << 20;
  Chunk slots[2];
  std::mutex mu;
  std::condition_variable cv;
  int filled[2] = {0, 0};  // 0 = free for the producer, 1 = ready for the consumer
  std::atomic<bool> abort{false};
  double parse_busy = 0;

  // ---- producer: inflate + parse into the staging slots ----
  std::thread producer([&]() {
    std::vector<char> buf(1 << 22);
    size_t have = 0, pos = 0;
    bool eof = false;
    std::string line;
    auto next_line = [&](std::string& out) -> bool {  // false at end of file; strips \r\n
      out.clear();
      for (;;) {
        if (pos == have) {
          if (eof) return !out.empty();
          const int n = gzread(gz, buf.data(), static_cast<unsigned>(buf.size()));
          if (n < 0 || (n == 0 && !gzeof(gz))) {  // Z_DATA_ERROR / Z_BUF_ERROR: a corrupt or truncated archive is not an end of file
            int zerr = 0;
            const char* zmsg = gzerror(gz, &zerr);
            throw std::invalid_argument(std::string("[bioparser] error: corrupt or truncated file (zlib: ") +
                                        (zmsg && *zmsg ? zmsg : "unexpected end") + ")");
          }
          if (n == 0) {
            // gzread returns 0 with gzeof() set also when the stream ends inside a member: zlib flags that case in gzerror
            int zerr = 0;
            (void)gzerror(gz, &zerr);
            if (zerr != Z_OK && zerr != Z_STREAM_END)
              throw std::invalid_argument("[bioparser] error: corrupt or truncated file (zlib: unexpected end of file)");
            eof = true;
            have = pos = 0;
            return !out.empty();
          }
          have = static_cast<size_t>(n);
          pos = 0;
        }
        const char* p = static_cast<const char*>(std::memchr(buf.data() + pos, '\n', have - pos));
        if (p) {
          out.append(buf.data() + pos, static_cast<size_t>(p - (buf.data() + pos)));
          pos = static_cast<size_t>(p - buf.data()) + 1;
          if (!out.empty() && out.back() == '\r') out.pop_back();
          return true;
        }
        out.append(buf.data() + pos, have - pos);
        pos = have;
      }
    };
    int cur = 0;
    Chunk* C = nullptr;
    auto acquire = [&]() {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return filled[cur] == 0 || abort.load(); });
      C = &slots[cur];
      C->n_chars = 0;
      C->lengths.clear();
      C->names.clear();
      C->last = false;
      C->error.clear();
    };
    auto publish = [&](bool last) {
      C->last = last;
      {
        std::lock_guard<std::mutex> lk(mu);
        filled[cur] = 1;
      }
      cv.notify_all();
      cur ^= 1;
    };
    std::string header, seq, qual;
    bool pending_header = false;
    try {
      acquire();
      auto emit = [&]() {  // one parsed record into the current chunk
        if (seq.size() > 0xFFFFFFFFULL) throw std::invalid_argument("[raven_hip] sequence longer than 2^32 bases");
        if (C->n_chars && C->n_chars + seq.size() > kChunkChars) {
          publish(false);
          acquire();
        }
        if (abort.load()) return;
        const u64 need = C->n_chars + seq.size();
        u8* dst = C->chars.get<u8>(std::max<u64>(need, kChunkChars));  // grows only for a read longer than a chunk
        std::memcpy(dst + C->n_chars, seq.data(), seq.size());
        if (fastq) {
          u8* dq = C->quals.get<u8>(std::max<u64>(need, kChunkChars));
          std::memcpy(dq + C->n_chars, qual.data(), qual.size());
        }
        C->n_chars = need;
        C->lengths.push_back(static_cast<u32>(seq.size()));
        size_t sp = 1;
        while (sp < header.size() && !std::isspace(static_cast<unsigned char>(header[sp]))) ++sp;
        C->names.emplace_back(header.substr(1, sp - 1));
      };
      const auto t0 = std::chrono::steady_clock::now();
      while (!abort.load()) {
        if (!pending_header) {
          if (!next_line(line)) break;
          if (line.empty()) continue;
          header = line;
        }
        pending_header = false;
        seq.clear();
        qual.clear();
        if (!fastq) {
          if (header[0] != '>') throw std::invalid_argument("[bioparser::FastaParser] error: invalid file format");
          std::string next_header;
          bool have_next = false;
          while (next_line(line)) {
            if (!line.empty() && line[0] == '>') {
              next_header = line;
              have_next = true;
              break;
            }
            seq += line;
          }
          emit();
          if (have_next) {
            header = next_header;
            pending_header = true;
          }
        } else {
          if (header[0] != '@') throw std::invalid_argument("[bioparser::FastqParser] error: invalid file format");
          bool plus = false;
          while (next_line(line)) {
            if (!line.empty() && line[0] == '+') {
              plus = true;
              break;
            }
            seq += line;
          }
          if (!plus) throw std::invalid_argument("[bioparser::FastqParser] error: invalid file format");
          while (qual.size() < seq.size() && next_line(line)) qual += line;
          if (qual.size() != seq.size()) throw std::invalid_argument("[bioparser::FastqParser] error: invalid file format");
          emit();
        }
      }
      parse_busy = secs(t0);
      if (!abort.load()) publish(true);
    } catch (const std::exception& ex) {
      if (C) {
        C->error = ex.what();
        C->n_chars = 0;
        C->lengths.clear();
        publish(true);
      }
    }
  });

  // ---- consumer: H2D + device packing ----
  struct Joiner {
    std::thread& t;
    std::atomic<bool>& abort;
    std::condition_variable& cv;
    gzFile gz;
    ~Joiner() {
      abort.store(true);
      cv.notify_all();
      if (t.joinable()) t.join();
      gzclose(gz);
    }
  } joiner{producer, abort, cv, gz};

  names.clear();
  R.h_len.clear();
  R.total_bases = 0;
  R.h_word_off.assign(1, 0);
  std::vector<u64> h_qoff(1, 0);
  u64 words_used = 0, qblocks_used = 0;
  DevBuf d_chars, d_qchars, d_boff, d_woff, d_qoff;
  u32* d_bad = e.tmp_f.get<u32>(4);
  RVN_HIP(hipMemsetAsync(d_bad, 0, 4, s));
  std::string error;
  double h2d_busy = 0;
  for (int cur = 0;; cur ^= 1) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return filled[cur] == 1; });
    }
    Chunk& C = slots[cur];
    if (!C.error.empty()) error = C.error;
    const u32 nr = static_cast<u32>(C.lengths.size());
    if (nr && error.empty()) {
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<u64> boff(nr + 1, 0), woff(nr + 1, 0), qoff(nr + 1, 0);
      for (u32 i = 0; i < nr; ++i) {
        boff[i + 1] = boff[i] + C.lengths[i];
        woff[i + 1] = woff[i] + (static_cast<u64>(C.lengths[i]) + 31) / 32;
        qoff[i + 1] = qoff[i] + (static_cast<u64>(C.lengths[i]) + 63) / 64;
      }
      const u64 nw = woff[nr], nq = qoff[nr];
      u8* dc = d_chars.get<u8>(C.n_chars + 16);
      u64* dbo = d_boff.get<u64>(nr + 1);
      u64* dwo = d_woff.get<u64>(nr + 1);
      RVN_HIP(hipMemcpyAsync(dc, C.chars.ptr, C.n_chars, hipMemcpyHostToDevice, s));
      RVN_HIP(hipMemcpyAsync(dbo, boff.data(), (nr + 1) * 8, hipMemcpyHostToDevice, s));
      RVN_HIP(hipMemcpyAsync(dwo, woff.data(), (nr + 1) * 8, hipMemcpyHostToDevice, s));
      ensure_capacity(R.packed, words_used * 8, (words_used + nw + 2) * 8, s);
      if (nw) {
        pack_ascii_kernel<<<div_up(nw, 256), 256, 0, s>>>(dc, dbo, dwo, nr, nw, R.packed.as<u64>() + words_used, d_bad);
        RVN_LAUNCH_CHECK();
      }
      if (fastq) {
        u8* dq = d_qchars.get<u8>(C.n_chars + 16);
        u64* dqo = d_qoff.get<u64>(nr + 1);
        RVN_HIP(hipMemcpyAsync(dq, C.quals.ptr, C.n_chars, hipMemcpyHostToDevice, s));
        RVN_HIP(hipMemcpyAsync(dqo, qoff.data(), (nr + 1) * 8, hipMemcpyHostToDevice, s));
        ensure_capacity(R.quals, qblocks_used, qblocks_used + nq + 16, s);
        if (nq) {
          block_quality_kernel<<<div_up(nq, 256), 256, 0, s>>>(dq, dbo, dqo, nr, nq, R.quals.as<u8>() + qblocks_used);
          RVN_LAUNCH_CHECK();
        }
      }
      RVN_HIP(rvn_stream_sync(s));  // the staging slot and the local offset arrays are free again
      for (u32 i = 0; i < nr; ++i) {
        R.h_len.push_back(C.lengths[i]);
        R.h_word_off.push_back(words_used + woff[i + 1]);
        h_qoff.push_back(qblocks_used + qoff[i + 1]);
        R.total_bases += C.lengths[i];
        names.emplace_back(std::move(C.names[i]));
      }
      words_used += nw;
      qblocks_used += nq;
      h2d_busy += secs(t0);
    }
    const bool last = C.last;
    {
      std::lock_guard<std::mutex> lk(mu);
      filled[cur] = 0;
    }
    cv.notify_all();
    if (last) break;
  }
  if (!error.empty()) throw std::invalid_argument(error);
  if (read_back(e, d_bad, 4) != 0)
    throw std::invalid_argument("[biosoup::NucleicAcid::NucleicAcid] error: not a nucleotide");
  // finish the read set exactly as rvn_reads_upload does
  const u32 n = static_cast<u32>(R.h_len.size());
  if (n >= (1u << 31)) throw std::invalid_argument("[raven_hip] more than 2^31 sequences");
  R.n = n;
  R.n_words = words_used;
  R.h_id.resize(n);
  for (u32 i = 0; i < n; ++i) R.h_id[i] = i;
  R.ids_are_indices = true;
  ensure_capacity(R.packed, words_used * 8, (words_used + 2) * 8, s);
  RVN_HIP(hipMemsetAsync(R.packed.as<u64>() + words_used, 0, 16, s));
  u64* d_wo = R.word_off.get<u64>(static_cast<size_t>(n) + 1);
  u32* d_len = R.len.get<u32>(static_cast<size_t>(n) + 1);
  u32* d_id = R.id.get<u32>(static_cast<size_t>(n) + 1);
  RVN_HIP(hipMemcpyAsync(d_wo, R.h_word_off.data(), (static_cast<size_t>(n) + 1) * 8, hipMemcpyHostToDevice, s));
  if (n) {
    RVN_HIP(hipMemcpyAsync(d_len, R.h_len.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, s));
    RVN_HIP(hipMemcpyAsync(d_id, R.h_id.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, s));
  }
  if (fastq) {
    u64* d_qo = R.qual_off.get<u64>(static_cast<size_t>(n) + 1);
    RVN_HIP(hipMemcpyAsync(d_qo, h_qoff.data(), (static_cast<size_t>(n) + 1) * 8, hipMemcpyHostToDevice, s));
    R.h_qual_off = h_qoff;
    R.qual_shift = 6;
  }
  RVN_HIP(rvn_stream_sync(s));
  reads_build_tiles(e, R);
  st.n_sequences = n;
  st.n_bases = R.total_bases;
  st.has_quality = fastq ? 1 : 0;
  st.parse_s = parse_busy;
  st.device_s = h2d_busy;
  st.total_s = secs(t_all);
}

}  // namespace rvn
