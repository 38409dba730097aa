// Lifetime test of the single-stream path of the input pipeline (raven_amd/csrc/io_text.h: fast_stream_all + helper
// threads), built with -fsanitize=address by tests/test_io_text.py (the helpers outlived the buffers they
// work on whenever the attempt ended early).  "early <path>": the source is destroyed after its first slab, five times;
// "full <path>": the file is read to its end (a truncated archive ends in SpeculationFailed / invalid_argument).  A
// use-after-free is AddressSanitizer's report on stderr and a non-zero exit code.
#include "io_text.h"
#include <cstdio>
#include <cstring>
int main(int argc, char** argv) {
  using namespace rvn::io;
  for (int a = 1; a + 1 < argc; a += 2) {
    const bool early = std::strcmp(argv[a], "early") == 0;
    for (int rep = 0; rep < (early ? 5 : 2); ++rep) {
      try {
        SourceOptions opt;
        opt.threads = 4;
        opt.slab_bytes = 1u << 20;
        TextSource src(argv[a + 1], opt);
        u8* slab;
        u64 n, total = 0;
        bool first = true;
        while (src.next(&slab, &n)) {
          total += n;
          if (early) break;  // the source goes out of scope with the decoder and its helpers in full flight
          if (!first) src.release();
          first = false;
        }
        std::printf("%s %s: fast %d, %llu bytes of text seen\n", argv[a], argv[a + 1], (int)src.fast_stream(), (unsigned long long)total);
      } catch (const SpeculationFailed&) {
        std::printf("%s %s: speculation failed\n", argv[a], argv[a + 1]);
      } catch (const std::invalid_argument& e) {
        std::printf("%s %s: error %s\n", argv[a], argv[a + 1], e.what());
      }
    }
  }
  return 0;
}
