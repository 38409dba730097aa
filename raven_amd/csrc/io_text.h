// io_text.h — host half of the input path (io.hip): a sequence file -> its TEXT in page-locked slabs -> records.
//
//   TextSource     the file is mapped; a gzip file is cut into its members (BGZF: from the BC extra field; any other
//                  multi-member archive: at every position that passes a strict gzip-header test) and the members are
//                  inflated by a pool of threads STRAIGHT INTO THEIR FINAL PLACE of the text (the ISIZE trailers give
//                  every member's place before anything is inflated).  The text lives in a ring of fixed-size slabs; the
//                  consumer gets slab 0, 1, 2 ... as each becomes complete.  The cut is a speculation and every member is
//                  verified (zlib checks CRC-32 and ISIZE; the stream must end exactly at the next cut): any mismatch
//                  makes the source report `speculation_failed`, and the caller starts over in streaming mode = one
//                  thread inflating the archive front to back (what gzread does; also the path of a single-member
//                  archive — one deflate stream cannot be entered in the middle).  Plain files: the pool copies ranges.
//   RecordScanner  one pass of memchr over a slab: FASTA / FASTQ records as bioparser delimits them (multi-line
//                  sequences and qualities, "\r\n", blank lines between records), producing per record the offsets of its
//                  bases and qualities IN THE KEPT TEXT.  The kept text is what goes to the device: the slab itself,
//                  untouched, while every field seen so far in the slab sits on one line (nothing is copied on the host);
//                  from the first continuation line on, the rest of the slab is closed up in place (memmove) so that a
//                  field is always one contiguous run.
//
// Restates what raven gets from bioparser::Parser<biosoup::NucleicAcid>::Parse(-1) (RavenLib/src/io.cc:7-41,
// RavenExe/src/main.cc:258-299); bioparser is not in the reference tree: record rules from SURVEY.md App. A.4.
#ifndef RVN_IO_TEXT_H_
#define RVN_IO_TEXT_H_

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "inflate_fast.h"

namespace rvn {
namespace io {

using u8 = std::uint8_t;
using u32 = std::uint32_t;
using u64 = std::uint64_t;

constexpr u64 kSlabMargin = 64;  // bytes in front of a slab's text (the scanner may put one carried byte there)

struct SpeculationFailed : std::runtime_error {
  SpeculationFailed() : std::runtime_error("member cut not confirmed") {}
};

// ---- gzip member headers ----------------------------------------------------------------------
// Length of the gzip header at p (0: not a complete, well-formed header); *bgzf_size = whole-member size from a BGZF
// 'BC' extra subfield (0 if absent).
inline u64 gz_header_len(const u8* p, u64 avail, u32* bgzf_size) {
  if (bgzf_size) *bgzf_size = 0;
  if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
  const u8 flg = p[3];
  u64 off = 10;
  if (flg & 4) {  // FEXTRA
    if (off + 2 > avail) return 0;
    const u64 xlen = p[off] | (static_cast<u64>(p[off + 1]) << 8);
    off += 2;
    if (off + xlen > avail) return 0;
    for (u64 q = off; q + 4 <= off + xlen;) {
      const u64 slen = p[q + 2] | (static_cast<u64>(p[q + 3]) << 8);
      if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= off + xlen && bgzf_size)
        *bgzf_size = (p[q + 4] | (static_cast<u32>(p[q + 5]) << 8)) + 1;
      q += 4 + slen;
    }
    off += xlen;
  }
  if (flg & 8) {  // FNAME
    while (off < avail && p[off]) ++off;
    if (++off > avail) return 0;
  }
  if (flg & 16) {  // FCOMMENT
    while (off < avail && p[off]) ++off;
    if (++off > avail) return 0;
  }
  if (flg & 2) off += 2;  // FHCRC
  return off + 8 <= avail ? off : 0;
}

// the strict test a speculative cut must pass: magic, deflate, no reserved flag, XFL in {0, 2, 4}, a known OS byte
inline bool gz_header_plausible(const u8* p, u64 avail) {
  if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return false;
  if (p[8] != 0 && p[8] != 2 && p[8] != 4) return false;
  if (p[9] > 13 && p[9] != 255) return false;
  return gz_header_len(p, avail, nullptr) != 0;
}

struct Member {
  u64 in_off, in_len;    // bytes of the file
  u64 out_off, out_len;  // its place in the text
};

struct SourceOptions {
  u32 threads = 0;              // 0: hardware_concurrency - 2, at most 32 (engine option io_threads)
  bool force_streaming = false;  // one zlib inflate thread, front to back (the authority on damaged archives)
  bool zlib_only = false;        // a single member by zlib from the first attempt on (engine option io_zlib)
  bool debug = false;            // decoder / helper timings to stderr
  u32 stream_helpers = 3;        // single-member archive: threads beside the decoder (CRC-32 + copy into the slabs)
  u64 slab_bytes = 8ULL << 20;   // (engine option io_slab_mb)
  u32 ring = 8;                  // (engine option io_ring)
  u64 item_bytes = 1ULL << 20;   // text per work item of the pool (BGZF blocks are 64 kB: grouped)
  // page-locked allocation (hipHostMalloc / hipHostFree in the product; malloc / free in the CPU test hook)
  std::function<void*(size_t)> alloc = [](size_t n) { return std::malloc(n); };
  std::function<void(void*)> release = [](void* p) { std::free(p); };
};

class TextSource {
 public:
  TextSource(const std::string& path, const SourceOptions& opt) : opt_(opt) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::invalid_argument("[bioparser::Parser::Create] error: unable to open file " + path);
    struct stat sb;
    if (::fstat(fd_, &sb) != 0 || !S_ISREG(sb.st_mode)) {
      ::close(fd_);
      throw std::invalid_argument("[bioparser::Parser::Create] error: unable to open file " + path);
    }
    size_ = static_cast<u64>(sb.st_size);
    if (size_) {
      void* m = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
      if (m == MAP_FAILED) {
        ::close(fd_);
        throw std::invalid_argument("[bioparser::Parser::Create] error: unable to map file " + path);
      }
      base_ = static_cast<const u8*>(m);
      ::madvise(m, size_, MADV_SEQUENTIAL);
    }
    try {
      plan();
      start();
    } catch (...) {
      shutdown();
      throw;
    }
  }
  TextSource(const TextSource&) = delete;
  TextSource& operator=(const TextSource&) = delete;
  ~TextSource() { shutdown(); }

  // The next slab of the text (blocks until it is complete): false after the last one.  *text has kSlabMargin writable
  // bytes in front of it.  Throws std::invalid_argument for a corrupt archive, SpeculationFailed if a cut was wrong.
  bool next(u8** text, u64* n) {
    std::unique_lock<std::mutex> lk(mu_);
    const u64 k = consumed_;
    cv_.wait(lk, [&] { return failed_ || slab_done(k) || (total_known_ && k >= n_slabs_); });
    if (failed_) rethrow();
    if (total_known_ && k >= n_slabs_) return false;
    Slab& s = slabs_[k % ring_];
    *text = s.text;
    *n = s.filled;
    ++consumed_;
    return true;
  }
  // the oldest slab handed out and not yet released may be overwritten
  void release() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      ++released_;
    }
    cv_.notify_all();
  }
  bool gzip() const { return gzip_; }
  bool streaming() const { return streaming_; }
  bool fast_stream() const { return fast_stream_; }
  u32 members() const { return static_cast<u32>(members_.size()); }
  u32 threads() const { return static_cast<u32>(workers_.size()); }

 private:
  struct Slab {
    void* raw = nullptr;
    u8* text = nullptr;
    u64 index = ~0ULL;  // which slab of the text the slot holds
    u64 filled = 0, expect = 0;
    bool done = false;
  };
  struct Item {
    u32 first, last;  // members [first, last)
  };

  bool slab_done(u64 k) const {
    const Slab& s = slabs_[k % ring_];
    return s.index == k && s.done;
  }
  void rethrow() {
    if (spec_failed_) throw SpeculationFailed();
    throw std::invalid_argument(error_);
  }

  // ---- what the file is made of ----
  void plan() {
    gzip_ = size_ >= 2 && base_[0] == 0x1f && base_[1] == 0x8b;
    streaming_ = false;
    if (!gzip_) {
      for (u64 off = 0; off < size_; off += kPlainPiece)
        members_.push_back(Member{off, std::min(kPlainPiece, size_ - off), off, std::min(kPlainPiece, size_ - off)});
      total_ = size_;
    } else if (opt_.force_streaming || !cut_members()) {
      members_.clear();
      streaming_ = true;
    }
    u32 want = opt_.threads;
    if (want == 0) {
      const u32 hw = std::max(1u, std::thread::hardware_concurrency());
      want = std::min(32u, hw > 3 ? hw - 2 : 1u);
    }
    if (streaming_) {
      // a single member (or a cut that did not work out): one deflate stream, one decoder.  Unless the caller asked for
      // zlib (force_streaming: the second attempt after anything went wrong) it is inflate_fast.h with a few helpers that
      // checksum and place what it produces
      fast_stream_ = !opt_.force_streaming && !opt_.zlib_only;
      n_threads_ = 1;
      total_known_ = false;
      slab_bytes_ = opt_.slab_bytes;
    } else {
      for (u32 i = 0; i < members_.size();) {
        u32 j = i + 1;
        u64 acc = members_[i].out_len;
        while (j < members_.size() && acc + members_[j].out_len <= opt_.item_bytes) acc += members_[j++].out_len;
        items_.push_back(Item{i, j});
        i = j;
      }
      n_threads_ = static_cast<u32>(std::max<size_t>(1, std::min<size_t>(want, items_.size())));
      total_known_ = true;
      slab_bytes_ = std::min<u64>(opt_.slab_bytes, std::max<u64>(1ULL << 20, ((total_ + (1ULL << 20) - 1) >> 20) << 20));
      n_slabs_ = (total_ + slab_bytes_ - 1) / slab_bytes_;
    }
  }

  // members of a gzip file without inflating anything; false: one member, or nothing to gain -> streaming
  bool cut_members() {
    u32 bsize = 0;
    if (gz_header_len(base_, size_, &bsize) == 0) return false;
    std::vector<u64> cuts;
    if (bsize) {  // BGZF: every block names its own size
      u64 off = 0;
      while (off < size_) {
        u32 bs = 0;
        if (gz_header_len(base_ + off, size_ - off, &bs) == 0 || bs == 0 || off + bs > size_) return false;
        cuts.push_back(off);
        off += bs;
      }
    } else {
      cuts.push_back(0);
      for (u64 off = 1; off + 18 <= size_;) {
        const u8* hit = static_cast<const u8*>(std::memchr(base_ + off, 0x1f, size_ - 18 - off + 1));
        if (!hit) break;
        off = static_cast<u64>(hit - base_);
        if (gz_header_plausible(hit, size_ - off)) cuts.push_back(off);
        ++off;
      }
      if (cuts.size() < 2) return false;
    }
    cuts.push_back(size_);
    u64 out = 0;
    members_.reserve(cuts.size() - 1);
    for (size_t i = 0; i + 1 < cuts.size(); ++i) {
      const u64 a = cuts[i], b = cuts[i + 1];
      if (b - a < 18) return false;
      const u64 isize = base_[b - 4] | (static_cast<u64>(base_[b - 3]) << 8) | (static_cast<u64>(base_[b - 2]) << 16) |
                        (static_cast<u64>(base_[b - 1]) << 24);
      if (isize > (b - a) * 1100 + 64) return false;  // deflate cannot expand beyond ~1032x: not a trailer (or > 4 GiB)
      members_.push_back(Member{a, b - a, out, isize});
      out += isize;
    }
    total_ = out;
    return true;
  }

  // ---- the ring ----
  void start() {
    const u32 n_alloc =
        total_known_ ? static_cast<u32>(std::min<u64>(opt_.ring, std::max<u64>(1, n_slabs_))) : std::min(opt_.ring, fast_stream_ ? 4u : 3u);
    ring_ = n_alloc;
    slabs_.resize(n_alloc);
    for (Slab& s : slabs_) {
      s.raw = opt_.alloc(slab_bytes_ + kSlabMargin + 64);
      if (!s.raw) throw std::bad_alloc();
      s.text = static_cast<u8*>(s.raw) + kSlabMargin;
    }
    for (u64 k = 0; k < n_alloc; ++k) assign(k);
    next_assign_ = n_alloc;
    for (u32 t = 0; t < n_threads_; ++t) workers_.emplace_back([this] { work(); });
  }
  void assign(u64 k) {  // slot k % ring now holds slab k (mu_ held, or before the workers start)
    Slab& s = slabs_[k % ring_];
    s.index = k;
    s.filled = 0;
    s.done = false;
    s.expect = total_known_ ? (k < n_slabs_ ? std::min(slab_bytes_, total_ - k * slab_bytes_) : 0) : slab_bytes_;
  }
  // a writer needs slab k: blocks until the slot is free of slab k - ring; nullptr when the source is being torn down
  Slab* writable(u64 k) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return stop_ || failed_ || k < released_ + ring_; });
    if (stop_ || failed_) return nullptr;
    while (next_assign_ <= k) assign(next_assign_++);
    return &slabs_[k % ring_];
  }
  void wrote(Slab* s, u64 n, bool close_now = false) {
    std::lock_guard<std::mutex> lk(mu_);
    s->filled += n;
    if (s->filled == s->expect || close_now) {
      s->done = true;
      cv_.notify_all();
    }
  }
  void fail(const std::string& msg, bool spec) {
    std::lock_guard<std::mutex> lk(mu_);
    if (!failed_) {
      failed_ = true;
      spec_failed_ = spec;
      error_ = msg;
    }
    cv_.notify_all();
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : workers_)
      if (t.joinable()) t.join();
    workers_.clear();
    for (Slab& s : slabs_)
      if (s.raw) opt_.release(s.raw);
    slabs_.clear();
    if (base_) ::munmap(const_cast<u8*>(base_), size_);
    base_ = nullptr;
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
  }

  // ---- the pool ----
  void work() {
    try {
      if (streaming_) {
        if (fast_stream_) fast_stream_all();
        else stream_all();
        return;
      }
      z_stream zs;
      std::memset(&zs, 0, sizeof(zs));
      bool z_open = false;
      if (gzip_) {
        if (inflateInit2(&zs, 15 + 16) != Z_OK) throw std::runtime_error("inflateInit2");
        z_open = true;
      }
      for (;;) {
        const size_t it = next_item_.fetch_add(1);
        if (it >= items_.size()) break;
        bool ok = true;
        for (u32 m = items_[it].first; ok && m < items_[it].last; ++m) ok = gzip_ ? inflate_member(zs, members_[m]) : copy_member(members_[m]);
        if (!ok) break;
      }
      if (z_open) inflateEnd(&zs);
    } catch (const std::exception& ex) {
      fail(std::string("[bioparser] error: ") + ex.what(), false);
    }
  }

  bool copy_member(const Member& m) {
    u64 pos = m.out_off, left = m.out_len;
    const u8* src = base_ + m.in_off;
    while (left) {
      Slab* s = writable(pos / slab_bytes_);
      if (!s) return false;
      const u64 at = pos % slab_bytes_, n = std::min(left, slab_bytes_ - at);
      std::memcpy(s->text + at, src, n);
      wrote(s, n);
      src += n;
      pos += n;
      left -= n;
    }
    return true;
  }

  // one member with a speculated extent: must inflate to exactly out_len bytes and end exactly at in_len
  bool inflate_member(z_stream& zs, const Member& m) {
    if (inflateReset(&zs) != Z_OK) throw std::runtime_error("inflateReset");
    const u8* in = base_ + m.in_off;
    u64 in_left = m.in_len, pos = m.out_off, out_left = m.out_len;
    zs.next_in = const_cast<Bytef*>(in);
    zs.avail_in = 0;
    u8 sink[8];
    for (;;) {
      if (zs.avail_in == 0 && in_left) {
        const u64 n = std::min<u64>(in_left, 1ULL << 30);
        zs.avail_in = static_cast<uInt>(n);
        in_left -= n;
      }
      Slab* s = nullptr;
      u64 room = 0;
      if (out_left) {
        s = writable(pos / slab_bytes_);
        if (!s) return false;
        const u64 at = pos % slab_bytes_;
        room = std::min<u64>(std::min(out_left, slab_bytes_ - at), 1ULL << 30);
        zs.next_out = s->text + at;
      } else {
        zs.next_out = sink;  // the stream must end without producing anything more
      }
      zs.avail_out = static_cast<uInt>(out_left ? room : sizeof(sink));
      const int rc = inflate(&zs, Z_NO_FLUSH);
      const u64 made = (out_left ? room : sizeof(sink)) - zs.avail_out;
      if (!out_left && made) {
        fail("", true);
        return false;
      }
      if (made) {
        wrote(s, made);
        pos += made;
        out_left -= made;
      }
      if (rc == Z_STREAM_END) {
        if (out_left || zs.avail_in || in_left) {
          fail("", true);
          return false;
        }
        return true;
      }
      if (rc != Z_OK && rc != Z_BUF_ERROR) {  // a damaged archive or a wrong cut: the streaming pass tells which
        fail("", true);
        return false;
      }
      if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_left == 0) {  // input exhausted before the end of the stream
        fail("", true);
        return false;
      }
    }
  }

  // ---- single stream, fast decoder -------------------------------------------------------------------------------
  // The decoder (this thread) fills one of two linear buffers (32 KB of history in front); what it produced is cut into
  // pieces that helper threads checksum (zlib crc32, combined in order at the member's end) and copy to their place in
  // the slabs while the decoder goes on in the other buffer.  Any doubt — a stream the decoder refuses, a CRC-32 or ISIZE
  // that does not match — ends the attempt as a failed speculation: the caller starts over with zlib, which reports what
  // is wrong with the archive (or reads it, should the doubt have been this decoder's fault).
  struct Piece {
    const u8* src = nullptr;
    u64 len = 0, text_off = 0;
    unsigned long crc = 0;
    bool done = false;
  };
  void helper_loop() {
    for (;;) {
      Piece* pc = nullptr;
      {
        std::unique_lock<std::mutex> lk(hmu_);
        hcv_.wait(lk, [&] { return hstop_ || !hqueue_.empty(); });
        if (hqueue_.empty()) return;
        pc = hqueue_.back();
        hqueue_.pop_back();
      }
      unsigned long c = crc32(0L, Z_NULL, 0);
      for (u64 at = 0; at < pc->len;) {  // (zlib takes 32-bit lengths)
        const u64 n = std::min<u64>(pc->len - at, 1ULL << 30);
        c = crc32(c, pc->src + at, static_cast<uInt>(n));
        at += n;
      }
      pc->crc = c;
      bool ok = true;
      for (u64 at = 0; ok && at < pc->len;) {
        const u64 pos = pc->text_off + at;
        Slab* sl = writable(pos / slab_bytes_);
        if (!sl) {
          ok = false;
          break;
        }
        const u64 in_slab = pos % slab_bytes_, n = std::min(pc->len - at, slab_bytes_ - in_slab);
        std::memcpy(sl->text + in_slab, pc->src + at, n);
        wrote(sl, n);
        at += n;
      }
      {
        std::lock_guard<std::mutex> lk(hmu_);
        pc->done = true;
      }
      hdone_.notify_all();
    }
  }
  void fast_stream_all() {
    constexpr u64 kHist = 32768, kBuf = 8ULL << 20;
    const u32 n_help = std::max(1u, opt_.stream_helpers);
    // (order of declaration = reverse order of destruction: the buffers and the piece lists the helpers work on are
    // declared BEFORE the guard that joins the helpers, so on every exit — an error, a stop request, a trailer that does
    // not match — the helpers have returned before anything they may still read or write goes out of scope)
    std::vector<u8> lin[2];
    for (auto& v : lin) v.resize(kHist + kBuf + FastInflate::kOutMargin + 64);
    std::vector<Piece> pieces[2];
    std::vector<std::thread> helpers;
    struct Stop {
      TextSource* t;
      std::vector<std::thread>* h;
      ~Stop() {
        {
          std::lock_guard<std::mutex> lk(t->hmu_);
          t->hstop_ = true;
          t->hqueue_.clear();  // pieces nobody has started are dropped; one in progress is finished before the join returns
        }
        t->hcv_.notify_all();
        for (std::thread& x : *h)
          if (x.joinable()) x.join();
      }
    } stop{this, &helpers};
    for (u32 i = 0; i < n_help; ++i) helpers.emplace_back([this] { helper_loop(); });
    auto wait_pieces = [&](std::vector<Piece>& ps) {
      std::unique_lock<std::mutex> lk(hmu_);
      hdone_.wait(lk, [&] {
        for (const Piece& x : ps)
          if (!x.done) return false;
        return true;
      });
    };
    u64 text_off = 0;  // text produced so far (all members)
    u64 in_pos = 0;
    int cur = 0;
    bool any_member = false;
    FastInflate dec;
    while (in_pos < size_) {
      u32 unused = 0;
      if (size_ - in_pos < 2 || base_[in_pos] != 0x1f || base_[in_pos + 1] != 0x8b) {
        if (!any_member) return fail("", true);
        break;  // trailing garbage after a complete member: ignored, as zlib does
      }
      const u64 hdr = gz_header_len(base_ + in_pos, size_ - in_pos, &unused);
      if (!hdr) return fail("", true);
      any_member = true;
      dec.reset(base_ + in_pos + hdr, base_ + size_);
      u8* bufp = lin[cur].data();
      const u8* valid_from = bufp + kHist;  // a member starts without history
      u8* o = bufp + kHist;
      unsigned long crc = crc32(0L, Z_NULL, 0);
      u64 member_len = 0;
      std::vector<std::pair<unsigned long, u64>> done_crcs;  // (crc, length) of the pieces, in text order
      for (;;) {
        const auto t_dec = std::chrono::steady_clock::now();
        const FastInflate::Status st = dec.run(valid_from, &o, bufp + lin[cur].size());
        dbg_decode_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dec).count();
        if (st == FastInflate::kError) return fail("", true);
        // what is new in this buffer: [bufp + kHist, o) -> pieces for the helpers
        const u8* from = bufp + kHist;
        const u64 fresh = static_cast<u64>(o - from);
        pieces[cur].clear();
        const u64 per = std::max<u64>(1ULL << 20, (fresh + n_help - 1) / n_help);
        for (u64 at = 0; at < fresh; at += per) {
          Piece pc;
          pc.src = from + at;
          pc.len = std::min(per, fresh - at);
          pc.text_off = text_off + at;
          pieces[cur].push_back(pc);
        }
        {
          std::lock_guard<std::mutex> lk(hmu_);
          for (Piece& pc : pieces[cur]) hqueue_.push_back(&pc);
        }
        hcv_.notify_all();
        text_off += fresh;
        member_len += fresh;
        // the other buffer: its pieces of the round before must be finished before it is written again
        const int other = cur ^ 1;
        const auto t_wait = std::chrono::steady_clock::now();
        wait_pieces(pieces[other]);
        dbg_wait_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
        for (const Piece& x : pieces[other]) done_crcs.emplace_back(x.crc, x.len);
        pieces[other].clear();
        if (stop_requested()) return;
        if (st == FastInflate::kStreamEnd) break;
        // history: the last 32 KB of what has been produced go in front of the other buffer
        u8* nb = lin[other].data();
        const u64 have = static_cast<u64>(o - valid_from);
        const u64 keep = std::min<u64>(kHist, have);
        std::memcpy(nb + kHist - keep, o - keep, keep);
        valid_from = nb + kHist - keep;
        o = nb + kHist;
        bufp = nb;
        cur = other;
      }
      // end of the member: every piece done, checksum and length against the trailer
      wait_pieces(pieces[cur]);
      for (const Piece& x : pieces[cur]) done_crcs.emplace_back(x.crc, x.len);
      pieces[cur].clear();
      for (const auto& c : done_crcs) crc = crc32_combine(crc, c.first, static_cast<z_off_t>(c.second));
      const u8* tr = dec.input_position();
      if (tr + 8 > base_ + size_) return fail("", true);
      const u64 want_crc = tr[0] | (static_cast<u64>(tr[1]) << 8) | (static_cast<u64>(tr[2]) << 16) | (static_cast<u64>(tr[3]) << 24);
      const u64 want_len = tr[4] | (static_cast<u64>(tr[5]) << 8) | (static_cast<u64>(tr[6]) << 16) | (static_cast<u64>(tr[7]) << 24);
      if (want_crc != (crc & 0xFFFFFFFFUL) || want_len != (member_len & 0xFFFFFFFFULL)) return fail("", true);
      in_pos = static_cast<u64>(tr + 8 - base_);
    }
    if (opt_.debug)
      std::fprintf(stderr, "[raven_hip] single stream: %.3f s decoding, %.3f s waiting for the helpers, %.1f MB of text\n", dbg_decode_s_,
                   dbg_wait_s_, text_off / 1e6);
    // close the text (every piece has been written: the waits above)
    {
      std::lock_guard<std::mutex> lk(mu_);
      total_ = text_off;
      n_slabs_ = (text_off + slab_bytes_ - 1) / slab_bytes_;
      total_known_ = true;
      if (n_slabs_) {
        const u64 k = n_slabs_ - 1;
        while (next_assign_ <= k) assign(next_assign_++);  // (cannot happen: its pieces were written) 
        Slab& s = slabs_[k % ring_];
        if (s.index == k) {
          s.expect = text_off - k * slab_bytes_;
          if (s.filled == s.expect) s.done = true;
        }
      }
    }
    cv_.notify_all();
  }
  bool stop_requested() {
    std::lock_guard<std::mutex> lk(mu_);
    return stop_ || failed_;
  }

  // the whole archive front to back on this thread (gzread's semantics: members back to back, garbage after the last
  // member ignored, a stream cut short or damaged is an error)
  void stream_all() {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) throw std::runtime_error("inflateInit2");
    struct Closer {
      z_stream* z;
      ~Closer() { inflateEnd(z); }
    } closer{&zs};
    u64 in_pos = 0, k = 0;
    Slab* s = writable(0);
    if (!s) return;
    u64 at = 0;
    bool in_member = false;
    auto corrupt = [&](const char* what) {
      fail(std::string("[bioparser] error: corrupt or truncated file (zlib: ") + what + ")", false);
    };
    for (;;) {
      if (zs.avail_in == 0 && in_pos < size_) {
        const u64 n = std::min<u64>(size_ - in_pos, 1ULL << 30);
        zs.next_in = const_cast<Bytef*>(base_ + in_pos);
        zs.avail_in = static_cast<uInt>(n);
        in_pos += n;
      }
      if (!in_member) {
        if (zs.avail_in == 0 && in_pos >= size_) break;  // clean end
        const u8* p = zs.next_in;
        const u64 avail = zs.avail_in + (size_ - in_pos);
        if (avail < 2 || p[0] != 0x1f || p[1] != 0x8b) break;  // trailing garbage after a complete member: ignored
        in_member = true;
      }
      if (at == slab_bytes_) {
        s = writable(++k);
        if (!s) return;
        at = 0;
      }
      const u64 room = std::min<u64>(slab_bytes_ - at, 1ULL << 30);
      zs.next_out = s->text + at;
      zs.avail_out = static_cast<uInt>(room);
      const int rc = inflate(&zs, Z_NO_FLUSH);
      const u64 made = room - zs.avail_out;
      if (made) {
        at += made;
        wrote(s, made);
      }
      if (rc == Z_STREAM_END) {
        in_member = false;
        if (inflateReset(&zs) != Z_OK) throw std::runtime_error("inflateReset");
        continue;
      }
      if (rc == Z_BUF_ERROR) {
        if (zs.avail_in == 0 && in_pos >= size_) return corrupt("unexpected end of file");
        continue;
      }
      if (rc != Z_OK) return corrupt(zs.msg ? zs.msg : "data error");
    }
    // close the text: the current slab is the last one
    {
      std::lock_guard<std::mutex> lk(mu_);
      s->done = true;
      s->expect = s->filled;
      total_known_ = true;
      n_slabs_ = s->filled ? k + 1 : k;
    }
    cv_.notify_all();
  }

  static constexpr u64 kPlainPiece = 1ULL << 20;
  SourceOptions opt_;
  int fd_ = -1;
  const u8* base_ = nullptr;
  u64 size_ = 0, total_ = 0, slab_bytes_ = 0, n_slabs_ = 0;
  bool gzip_ = false, streaming_ = false, total_known_ = false, fast_stream_ = false;
  u32 n_threads_ = 1, ring_ = 1;
  std::vector<Member> members_;
  std::vector<Item> items_;
  std::vector<Slab> slabs_;
  std::vector<std::thread> workers_;
  std::atomic<size_t> next_item_{0};
  std::mutex mu_;
  std::condition_variable cv_;
  std::mutex hmu_;  // single-stream helpers: queue of pieces
  std::condition_variable hcv_, hdone_;
  std::vector<Piece*> hqueue_;
  bool hstop_ = false;
  double dbg_decode_s_ = 0, dbg_wait_s_ = 0;
  u64 consumed_ = 0, released_ = 0, next_assign_ = 0;
  bool stop_ = false, failed_ = false, spec_failed_ = false;
  std::string error_;
};

// ---- records ------------------------------------------------------------------------------------
struct TextRecord {
  u64 seq_off = 0, qual_off = 0;  // offsets in the kept text
  u64 len = 0;
};

class RecordScanner {
 public:
  explicit RecordScanner(bool fastq) : fastq_(fastq) {}

  // One slab (in order).  On return the kept bytes of the slab are [*run, *run + *run_len) and belong at offset
  // *run_base of the kept text (*run_base may lie up to two bytes before the end of the previous run: those bytes are
  // overwritten).  Records completed by this slab are appended to `records` / `names`.
  void scan(u8* text, u64 n, const u8** run, u64* run_len, u64* run_base, std::vector<TextRecord>& records,
            std::vector<std::string>& names) {
    u8* r = text;
    u8* const end = text + n;
    run_begin_ = text;
    w_ = text;
    run_base_ = text_end_;
    if (pending_cr_) {  // the previous slab ended in a '\r' of a field line
      pending_cr_ = false;
      if (r < end && *r == '\n') {
        // it closed the line: not data
      } else {
        *--run_begin_ = '\r';  // data after all (the slab's margin takes it)
        field_len() += 1;
        last_is_cr_ = false;
      }
    }
    while (r < end) {
      if (!mid_line_) {
        const u8 c = *r;
        if (state_ == kHdr) {
          if (c == '\n') {
            skip(r, 1);
            ++r;
            continue;
          }
          if (c == '\r') {
            kind_ = kBlank;
          } else {
            if (c != (fastq_ ? '@' : '>'))
              throw std::invalid_argument(fastq_ ? "[bioparser::FastqParser] error: invalid file format"
                                                 : "[bioparser::FastaParser] error: invalid file format");
            kind_ = kHdrLine;
            name_.clear();
            name_done_ = false;
            cur_ = TextRecord();
            seq_lines_ = qual_lines_ = 0;
            in_record_ = true;
            skip(r, 1);
            ++r;
          }
        } else if (state_ == kSeq) {
          if (fastq_ && c == '+') {
            kind_ = kPlusLine;
          } else if (!fastq_ && c == '>') {
            finish_record(records, names);
            state_ = kHdr;
            continue;
          } else {
            kind_ = kSeqLine;
            begin_field_line(cur_.seq_off, seq_lines_);
          }
        } else {  // kQual
          kind_ = kQualLine;
          begin_field_line(cur_.qual_off, qual_lines_);
        }
        mid_line_ = true;
        last_is_cr_ = false;
        if (r == end) break;
      }
      const u8* nl = static_cast<const u8*>(std::memchr(r, '\n', static_cast<size_t>(end - r)));
      u8* e = nl ? const_cast<u8*>(nl) : end;
      const u64 len = static_cast<u64>(e - r);
      switch (kind_) {
        case kHdrLine:
          if (!name_done_) {
            u64 x = 0;
            while (x < len && !is_space(r[x])) ++x;
            name_.append(reinterpret_cast<const char*>(r), x);
            if (x < len) name_done_ = true;
          }
          skip(r, len);
          break;
        case kBlank:
          for (u64 x = 0; x < len; ++x)
            if (r[x] != '\r')
              throw std::invalid_argument(fastq_ ? "[bioparser::FastqParser] error: invalid file format"
                                                 : "[bioparser::FastaParser] error: invalid file format");
          skip(r, len);
          break;
        case kPlusLine:
          skip(r, len);
          break;
        default: {  // a line of bases or qualities
          u64 keep_len = len;
          if (len) {
            last_is_cr_ = e[-1] == '\r';
            if (!nl && last_is_cr_) {  // '\r' at the very end of the slab: line end or data? the next slab tells
              pending_cr_ = true;
              last_is_cr_ = false;  // held back: neither kept nor counted yet
              keep_len = len - 1;
            }
          }
          keep(r, keep_len);
          field_len() += keep_len;
          tail_junk_ = 0;
          break;
        }
      }
      r = e;
      if (!nl) break;
      // ---- end of line ----
      if (kind_ == kSeqLine || kind_ == kQualLine) {
        if (last_is_cr_) {
          field_len() -= 1;
          tail_junk_ = 1;
        }
      }
      tail_junk_ += skip(r, 1);
      ++r;
      mid_line_ = false;
      end_of_line(records, names);
    }
    *run = run_begin_;
    *run_len = static_cast<u64>(w_ - run_begin_);
    *run_base = run_base_;
    text_end_ = run_base_ + *run_len;
  }

  // end of the text: the last record (bioparser: a FASTA record ends with the file; a FASTQ record must be complete).
  // Returns true if one more byte (*extra) belongs at the end of the kept text (a '\r' that ended the file inside a field).
  bool finish(std::vector<TextRecord>& records, std::vector<std::string>& names, u8* extra) {
    bool more = false;
    if (pending_cr_) {
      pending_cr_ = false;
      *extra = '\r';
      field_len() += 1;
      text_end_ += 1;
      more = true;
    }
    if (mid_line_) {
      mid_line_ = false;
      end_of_line(records, names);
    }
    if (!fastq_) {
      if (in_record_) finish_record(records, names);
    } else if (in_record_) {
      throw std::invalid_argument("[bioparser::FastqParser] error: invalid file format");
    }
    return more;
  }
  // where the kept text must be retained from: the start of the record in progress
  u64 retain_from() const { return in_record_ && seq_lines_ ? cur_.seq_off : text_end_; }
  u64 text_end() const { return text_end_; }

 private:
  enum State { kHdr, kSeq, kQual };
  enum Kind { kHdrLine, kBlank, kSeqLine, kPlusLine, kQualLine };

  static bool is_space(u8 c) { return c == ' ' || (c >= 9 && c <= 13); }
  u64 pos() const { return run_base_ + static_cast<u64>(w_ - run_begin_); }
  u64& field_len() { return kind_ == kQualLine ? qual_len_ : cur_.len; }

  // bytes that need not survive: kept as they are while the slab is still in place (free), dropped once it is shifted
  u64 skip(const u8* r, u64 n) {
    if (w_ == r) {
      w_ += n;
      return n;
    }
    return 0;
  }
  void keep(const u8* r, u64 n) {
    if (w_ != r && n) std::memmove(w_, r, n);
    w_ += n;
  }
  void truncate(u64 k) {
    const u64 have = static_cast<u64>(w_ - run_begin_);
    if (k <= have) {
      w_ -= k;
    } else {
      run_base_ -= k - have;
      w_ = run_begin_;
    }
  }
  // a line of the field starts: the first one fixes the field's offset, a later one must follow the last base directly
  void begin_field_line(u64& off, u32& lines) {
    if (lines == 0) {
      off = pos();
    } else {
      truncate(tail_junk_);
    }
    tail_junk_ = 0;
    ++lines;
  }
  void end_of_line(std::vector<TextRecord>& records, std::vector<std::string>& names) {
    switch (kind_) {
      case kHdrLine:
        state_ = kSeq;
        break;
      case kPlusLine:
        if (cur_.len == 0) {
          finish_record(records, names);
          state_ = kHdr;
        } else {
          state_ = kQual;
          qual_len_ = 0;
        }
        break;
      case kQualLine:
        if (qual_len_ > cur_.len) throw std::invalid_argument("[bioparser::FastqParser] error: invalid file format");
        if (qual_len_ == cur_.len) {
          finish_record(records, names);
          state_ = kHdr;
        }
        break;
      default:
        break;
    }
  }
  void finish_record(std::vector<TextRecord>& records, std::vector<std::string>& names) {
    if (cur_.len > 0xFFFFFFFFULL) throw std::invalid_argument("[raven_hip] sequence longer than 2^32 bases");
    records.push_back(cur_);
    names.emplace_back(std::move(name_));
    name_.clear();
    in_record_ = false;
  }

  const bool fastq_;
  State state_ = kHdr;
  Kind kind_ = kBlank;
  bool mid_line_ = false, last_is_cr_ = false, pending_cr_ = false, name_done_ = false, in_record_ = false;
  TextRecord cur_;
  u64 qual_len_ = 0, tail_junk_ = 0;
  u32 seq_lines_ = 0, qual_lines_ = 0;
  std::string name_;
  u8* run_begin_ = nullptr;
  u8* w_ = nullptr;
  u64 run_base_ = 0, text_end_ = 0;
};

}  // namespace io
}  // namespace rvn

#endif  // RVN_IO_TEXT_H_
