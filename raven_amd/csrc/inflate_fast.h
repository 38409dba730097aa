// inflate_fast.h — a table-driven raw-deflate decoder (RFC 1951) for the ONE case the inflate pool of io_text.h cannot
// help: an archive that is a single gzip member is one deflate stream and has to be decoded front to back by one thread,
// so that thread's speed is the input path's (zlib 1.2.11, the only inflate in the image: ~0.55 GB/s of text on FASTQ).
// Same technique as the fast decoders in the field (one 64-bit bit buffer refilled with an unaligned load, an 11-bit
// primary table for literals / lengths with subtables behind it, entries that carry base value, extra-bit count and code
// length, several literals per refill, word-wise match copies) — written for this repository, no code taken from any.
//
// The caller owns input and output: the whole compressed stream is in memory; the output is a contiguous buffer in which
// the bytes in front of the write position are the history a match may reach back into (up to 32 KB).  run() decodes until
// the final block ends (kStreamEnd), the output buffer is nearly full (kOutputFull: the caller drains it, keeps the last
// 32 KB in front and calls again) or the stream is invalid (kError).  Checksums are the caller's business.
#ifndef RVN_INFLATE_FAST_H_
#define RVN_INFLATE_FAST_H_

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace rvn {
namespace io {

class FastInflate {
 public:
  enum Status { kStreamEnd, kOutputFull, kError };
  static constexpr size_t kOutMargin = 320;  // run() stops while at least this much output room is left

  void reset(const std::uint8_t* in, const std::uint8_t* in_end) {
    in_ = in;
    in_end_ = in_end;
    bitbuf_ = 0;
    bitcnt_ = 0;
    state_ = kBlockHeader;
    final_ = false;
    error_ = nullptr;
  }
  const char* error() const { return error_; }
  // first input byte not consumed (whole bytes still in the bit buffer are given back)
  const std::uint8_t* input_position() const { return in_ - (bitcnt_ >> 3); }

  // out_begin: start of the valid history; *out: write position (advanced); out_end: end of the buffer
  Status run(const std::uint8_t* out_begin, std::uint8_t** out, std::uint8_t* out_end) {
    std::uint8_t* o = *out;
    for (;;) {
      if (state_ == kBlockHeader) {
        if (final_) {
          *out = o;
          return kStreamEnd;
        }
        if (!block_header()) return fail(out, o);
      }
      if (state_ == kStored) {
        while (stored_left_) {
          if (static_cast<size_t>(out_end - o) < kOutMargin) {
            *out = o;
            return kOutputFull;
          }
          size_t n = stored_left_;
          const size_t room = static_cast<size_t>(out_end - o), avail = static_cast<size_t>(in_end_ - in_);
          if (n > room) n = room;
          if (n > avail) n = avail;
          if (n == 0) {
            error_ = "unexpected end of file";
            return fail(out, o);
          }
          std::memcpy(o, in_, n);
          o += n;
          in_ += n;
          stored_left_ -= static_cast<std::uint32_t>(n);
        }
        state_ = kBlockHeader;
        continue;
      }
      // ---- a Huffman block ----
      for (;;) {
        if (static_cast<size_t>(out_end - o) < kOutMargin) {
          *out = o;
          return kOutputFull;
        }
        if (bitcnt_ > 64) {  // (a consume below zero: the input ended inside a symbol)
          error_ = "unexpected end of file";
          return fail(out, o);
        }
        refill();
        std::uint32_t e = lit_[bitbuf_ & kLitMask];
        if (e & kSub) e = lit_[(e >> 16) + ((bitbuf_ >> kLitBits) & ((1u << ((e >> 8) & 15)) - 1))];
        if (e & kLiteral) {  // up to three literals per refill (a literal code is at most 15 bits)
          consume(e & 15);
          *o++ = static_cast<std::uint8_t>(e >> 16);
          e = lit_[bitbuf_ & kLitMask];
          if (e & kSub) e = lit_[(e >> 16) + ((bitbuf_ >> kLitBits) & ((1u << ((e >> 8) & 15)) - 1))];
          if (!(e & kLiteral)) goto not_literal;
          consume(e & 15);
          *o++ = static_cast<std::uint8_t>(e >> 16);
          e = lit_[bitbuf_ & kLitMask];
          if (e & kSub) e = lit_[(e >> 16) + ((bitbuf_ >> kLitBits) & ((1u << ((e >> 8) & 15)) - 1))];
          if (!(e & kLiteral)) goto not_literal;
          consume(e & 15);
          *o++ = static_cast<std::uint8_t>(e >> 16);
          continue;
        }
      not_literal:
        if (bitcnt_ > 64) {
          error_ = "unexpected end of file";
          return fail(out, o);
        }
        if (bitcnt_ < 48) refill();  // a length (<= 15 + 5 bits) and a distance (<= 15 + 13 bits) follow
        if (e & kEndOfBlock) {
          if (bitcnt_ < (e & 15)) {
            error_ = "unexpected end of file";
            return fail(out, o);
          }
          consume(e & 15);
          state_ = kBlockHeader;
          break;
        }
        if (!(e & kLength)) {
          error_ = "invalid literal/length code";
          return fail(out, o);
        }
        consume(e & 15);
        std::uint32_t len = (e >> 16) + static_cast<std::uint32_t>(bitbuf_ & ((1u << ((e >> 8) & 15)) - 1));
        consume((e >> 8) & 15);
        std::uint32_t d = dist_[bitbuf_ & kDistMask];
        if (d & kSub) d = dist_[(d >> 16) + ((bitbuf_ >> kDistBits) & ((1u << ((d >> 8) & 15)) - 1))];
        if (!(d & kLength)) {
          error_ = "invalid distance code";
          return fail(out, o);
        }
        consume(d & 15);
        const std::uint32_t dist = (d >> 16) + static_cast<std::uint32_t>(bitbuf_ & ((1u << ((d >> 8) & 15)) - 1));
        consume((d >> 8) & 15);
        if (bitcnt_ > 64) {  // (consume went below zero: the input ended inside the symbol)
          error_ = "unexpected end of file";
          return fail(out, o);
        }
        if (dist > static_cast<size_t>(o - out_begin)) {
          error_ = "invalid distance too far back";
          return fail(out, o);
        }
        // the copy: at least kOutMargin (> 258 + 16) bytes of room, so whole words may run past the match's end
        const std::uint8_t* s = o - dist;
        std::uint8_t* const end = o + len;
        if (dist >= 8) {
          do {
            std::memcpy(o, s, 8);
            o += 8;
            s += 8;
          } while (o < end);
        } else if (dist == 1) {
          std::memset(o, *s, len);
        } else {
          do {
            *o++ = *s++;
          } while (o < end);
        }
        o = end;
      }
    }
  }

 private:
  enum State { kBlockHeader, kStored, kHuffman };
  static constexpr int kLitBits = 11, kDistBits = 8;
  static constexpr std::uint32_t kLitMask = (1u << kLitBits) - 1, kDistMask = (1u << kDistBits) - 1;
  // table entry: bits 0-3 code length (primary) or remaining length (sub), 8-11 extra bits (or subtable index bits),
  // 16-31 literal / base value / subtable offset; flags:
  static constexpr std::uint32_t kLiteral = 1u << 4, kLength = 1u << 5, kEndOfBlock = 1u << 6, kSub = 1u << 7;

  Status fail(std::uint8_t** out, std::uint8_t* o) {
    *out = o;
    if (!error_) error_ = "invalid block";
    return kError;
  }
  void refill() {
    if (in_end_ - in_ >= 8) {
      std::uint64_t w;
      std::memcpy(&w, in_, 8);
      bitbuf_ |= w << bitcnt_;
      in_ += (63 - bitcnt_) >> 3;
      bitcnt_ |= 56;
    } else {
      while (bitcnt_ <= 56 && in_ < in_end_) {
        bitbuf_ |= static_cast<std::uint64_t>(*in_++) << bitcnt_;
        bitcnt_ += 8;
      }
    }
  }
  // (bitcnt_ is unsigned: consuming more bits than there are wraps it far above 64, which the callers test for)
  void consume(std::uint32_t n) {
    bitbuf_ >>= n;
    bitcnt_ -= n;
  }
  bool need(std::uint32_t n) {
    if (bitcnt_ < n) refill();
    if (bitcnt_ < n || bitcnt_ > 64) {
      error_ = "unexpected end of file";
      return false;
    }
    return true;
  }
  std::uint32_t take(std::uint32_t n) {
    const std::uint32_t v = static_cast<std::uint32_t>(bitbuf_ & ((1ULL << n) - 1));
    consume(n);
    return v;
  }

  bool block_header() {
    if (!need(3)) return false;
    final_ = take(1) != 0;
    const std::uint32_t type = take(2);
    if (type == 0) {
      // to the next byte boundary; LEN and NLEN follow
      consume(bitcnt_ & 7);
      if (!need(32)) return false;
      const std::uint32_t len = take(16), nlen = take(16);
      if ((len ^ 0xFFFFu) != nlen) {
        error_ = "invalid stored block lengths";
        return false;
      }
      // the bit buffer holds whole bytes now: give them back to the byte stream
      in_ -= bitcnt_ >> 3;
      bitbuf_ = 0;
      bitcnt_ = 0;
      stored_left_ = len;
      state_ = kStored;
      return true;
    }
    if (type == 1) {
      std::uint8_t lens[320];
      for (int i = 0; i < 144; ++i) lens[i] = 8;
      for (int i = 144; i < 256; ++i) lens[i] = 9;
      for (int i = 256; i < 280; ++i) lens[i] = 7;
      for (int i = 280; i < 288; ++i) lens[i] = 8;
      for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
      if (!build(lens, 288, lit_, kLitBits, true) || !build(lens + 288, 32, dist_, kDistBits, false)) return false;
      state_ = kHuffman;
      return true;
    }
    if (type == 3) {
      error_ = "invalid block type";
      return false;
    }
    // dynamic code
    if (!need(14)) return false;
    const std::uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
    if (hlit > 286 || hdist > 30) {
      error_ = "too many length or distance symbols";
      return false;
    }
    static const std::uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    std::uint8_t cl[19] = {};
    for (std::uint32_t i = 0; i < hclen; ++i) {
      if (!need(3)) return false;
      cl[order[i]] = static_cast<std::uint8_t>(take(3));
    }
    std::uint32_t pre[1 << 7];
    if (!build(cl, 19, pre, 7, false, true)) return false;
    std::uint8_t lens[320] = {};
    for (std::uint32_t i = 0; i < hlit + hdist;) {
      if (!need(7 + 7)) return false;
      const std::uint32_t e = pre[bitbuf_ & 127];
      if (!(e & kLength)) {
        error_ = "invalid code lengths set";
        return false;
      }
      consume(e & 15);
      const std::uint32_t sym = e >> 16;
      if (sym < 16) {
        lens[i++] = static_cast<std::uint8_t>(sym);
        continue;
      }
      std::uint32_t rep, val = 0;
      if (sym == 16) {
        if (i == 0) {
          error_ = "invalid bit length repeat";
          return false;
        }
        val = lens[i - 1];
        rep = 3 + take(2);
      } else if (sym == 17) {
        rep = 3 + take(3);
      } else {
        rep = 11 + take(7);
      }
      if (i + rep > hlit + hdist) {
        error_ = "invalid bit length repeat";
        return false;
      }
      while (rep--) lens[i++] = static_cast<std::uint8_t>(val);
    }
    if (bitcnt_ > 64) {
      error_ = "unexpected end of file";
      return false;
    }
    if (lens[256] == 0) {
      error_ = "invalid code -- missing end-of-block";
      return false;
    }
    std::uint8_t dl[32] = {};
    std::memcpy(dl, lens + hlit, hdist);
    if (!build(lens, hlit, lit_, kLitBits, true) || !build(dl, hdist, dist_, kDistBits, false)) return false;
    state_ = kHuffman;
    return true;
  }

  // Canonical Huffman code -> decoding table.  litlen: symbols are literals / end of block / lengths; otherwise
  // distances, or (plain) the code-length alphabet, whose entries carry the symbol itself.
  bool build(const std::uint8_t* lens, int n, std::uint32_t* tab, int bits, bool litlen, bool plain = false) {
    static const std::uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const std::uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const std::uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const std::uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int count[16] = {};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int left = 1, max_len = 0, used = 0;
    for (int l = 1; l <= 15; ++l) {
      left = (left << 1) - count[l];
      if (left < 0) {
        error_ = "invalid code lengths set (over-subscribed)";
        return false;
      }
      if (count[l]) max_len = l;
      used += count[l];
    }
    // an incomplete code is an error, except (as in zlib) a literal/length or distance code whose longest code has one bit,
    // and an empty distance alphabet (a block of literals only: meeting a distance there is the error)
    if (left > 0 && (plain || max_len > 1) && !(used == 0 && !litlen && !plain)) {
      error_ = "invalid code lengths set (incomplete)";
      return false;
    }
    const std::uint32_t primary = 1u << bits;
    for (std::uint32_t i = 0; i < primary; ++i) tab[i] = 0;  // 0 = no flag: an invalid code
    std::uint32_t next_code[16];
    {
      std::uint32_t code = 0;
      for (int l = 1; l <= 15; ++l) {
        code = (code + static_cast<std::uint32_t>(count[l - 1])) << 1;
        next_code[l] = code;
      }
    }
    auto entry_of = [&](int sym, int len_field) -> std::uint32_t {
      if (plain) return (static_cast<std::uint32_t>(sym) << 16) | kLength | static_cast<std::uint32_t>(len_field);
      if (litlen) {
        if (sym < 256) return (static_cast<std::uint32_t>(sym) << 16) | kLiteral | static_cast<std::uint32_t>(len_field);
        if (sym == 256) return kEndOfBlock | static_cast<std::uint32_t>(len_field);
        if (sym > 285) return static_cast<std::uint32_t>(len_field);  // 286, 287: no flag = invalid when met
        return (static_cast<std::uint32_t>(lbase[sym - 257]) << 16) | (static_cast<std::uint32_t>(lext[sym - 257]) << 8) | kLength |
               static_cast<std::uint32_t>(len_field);
      }
      if (sym > 29) return static_cast<std::uint32_t>(len_field);
      return (static_cast<std::uint32_t>(dbase[sym]) << 16) | (static_cast<std::uint32_t>(dext[sym]) << 8) | kLength |
             static_cast<std::uint32_t>(len_field);
    };
    auto reverse = [](std::uint32_t code, int len) {
      std::uint32_t r = 0;
      for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
      return r;
    };
    // subtables: one per distinct primary prefix of the codes longer than `bits`, sized for the longest code behind it
    std::uint32_t sub_next = primary;
    const int cap = litlen ? kLitTable : (plain ? 128 : kDistTable);
    // pass 1: codes that fit the primary table; pass 2: the long ones (they come last in canonical order)
    for (int pass = 0; pass < 2; ++pass) {
      std::uint32_t nc[16];
      std::memcpy(nc, next_code, sizeof(nc));
      for (int sym = 0; sym < n; ++sym) {
        const int l = lens[sym];
        if (l == 0) continue;
        const std::uint32_t code = nc[l]++;
        const std::uint32_t rev = reverse(code, l);
        if (l <= bits) {
          if (pass) continue;
          const std::uint32_t e = entry_of(sym, l);
          for (std::uint32_t i = rev; i < primary; i += 1u << l) tab[i] = e;
        } else {
          if (!pass) continue;
          const std::uint32_t prefix = rev & (primary - 1);
          if (!(tab[prefix] & kSub)) {
            // width of this subtable: the longest code sharing the prefix (codes of one prefix are consecutive in
            // canonical order, so scanning the symbols from here on finds them all)
            int longest = l;
            {
              std::uint32_t nc2[16];
              std::memcpy(nc2, nc, sizeof(nc2));
              nc2[l]--;  // this symbol again
              for (int s2 = sym; s2 < n; ++s2) {
                const int l2 = lens[s2];
                if (l2 <= bits) continue;
                const std::uint32_t r2 = reverse(nc2[l2]++, l2);
                if ((r2 & (primary - 1)) == prefix && l2 > longest) longest = l2;
              }
            }
            const int sb = longest - bits;
            if (sub_next + (1u << sb) > static_cast<std::uint32_t>(cap)) {
              error_ = "invalid code lengths set (table overflow)";
              return false;
            }
            tab[prefix] = (sub_next << 16) | (static_cast<std::uint32_t>(sb) << 8) | kSub | static_cast<std::uint32_t>(bits);
            for (std::uint32_t i = 0; i < (1u << sb); ++i) tab[sub_next + i] = 0;
            sub_next += 1u << sb;
          }
          const std::uint32_t base = tab[prefix] >> 16, sb = (tab[prefix] >> 8) & 15;
          const std::uint32_t e = entry_of(sym, l);  // the whole code length: primary bits + what the subtable consumed
          for (std::uint32_t i = rev >> bits; i < (1u << sb); i += 1u << (l - bits)) tab[base + i] = e;
        }
      }
    }
    return true;
  }

  static constexpr int kLitTable = 2048 + 2048, kDistTable = 256 + 1024;
  const std::uint8_t* in_ = nullptr;
  const std::uint8_t* in_end_ = nullptr;
  std::uint64_t bitbuf_ = 0;
  std::uint32_t bitcnt_ = 0;
  State state_ = kBlockHeader;
  bool final_ = false;
  std::uint32_t stored_left_ = 0;
  const char* error_ = nullptr;
  std::uint32_t lit_[kLitTable];
  std::uint32_t dist_[kDistTable];
};

}  // namespace io
}  // namespace rvn

#endif  // RVN_INFLATE_FAST_H_
