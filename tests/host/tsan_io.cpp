// The host half of the input path (raven_amd/csrc/io_text.h: inflate pool, single-stream decoder + helpers, slab ring,
// record scanner) in a program of its own, built with -fsanitize=thread by tests/test_io_text.py: every file given as
// "<q|a> <path>" (FASTQ / FASTA) is read with slabs of 8 MB, 70 001 and 1 031 bytes; a data race is ThreadSanitizer's
// report on stderr and a non-zero exit code.
#include "io_text.h"
#include <cstdio>
int main(int argc, char** argv) {
  using namespace rvn::io;
  for (int a = 1; a + 1 < argc; a += 2) {
    const bool fastq = argv[a][0] == 'q';
    for (int rep = 0; rep < 3; ++rep) {
      for (int attempt = 0; attempt < 2; ++attempt) {
        try {
          SourceOptions opt;
          opt.threads = 4;
          opt.slab_bytes = rep == 0 ? (8u << 20) : (rep == 1 ? 70001 : 1031);
          opt.force_streaming = attempt == 1;
          TextSource src(argv[a + 1], opt);
          RecordScanner sc(fastq);
          std::vector<TextRecord> recs; std::vector<std::string> nm;
          u8* slab; u64 n; bool first = true; u64 total = 0;
          while (src.next(&slab, &n)) {
            const u8* run; u64 rl, rb;
            sc.scan(slab, n, &run, &rl, &rb, recs, nm);
            total += n;
            if (!first) src.release();
            first = false;
          }
          u8 extra;
          sc.finish(recs, nm, &extra);
          std::printf("%s slab %llu streaming %d fast %d: %zu records, %llu bytes of text\n", argv[a + 1], (unsigned long long)opt.slab_bytes,
                      (int)src.streaming(), (int)src.fast_stream(), recs.size(), (unsigned long long)total);
          break;
        } catch (const SpeculationFailed&) { std::printf("restart\n"); }
      }
    }
  }
  return 0;
}
