// tools/real_deps/ref_harness.cpp — built ONLY by tools/fetch_real_deps.sh against the real ram / edlib / racon
// (never in this repository's normal build; the libraries are absent from the container).  Prints what
// tools/real_deps/compare_with_oracle.py diffs with oracle/:
//   ref_harness map     <reads.fastq.gz> <k> <w> <freq> <minhash 0|1>   -> "occ <occurrence>" then one line per overlap
//   ref_harness edlib   <reads.fastq.gz> <n_pairs>                       -> one distance per line (read i vs read i + 1)
//   ref_harness polish  <reads.fastq.gz> <target.fasta.gz>               -> one consensus per line
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "bioparser/fasta_parser.hpp"
#include "bioparser/fastq_parser.hpp"
#include "biosoup/nucleic_acid.hpp"
#include "edlib.h"  // NOLINT
#include "racon/polisher.hpp"
#include "ram/minimizer_engine.hpp"
#include "thread_pool/thread_pool.hpp"

std::atomic<std::uint32_t> biosoup::NucleicAcid::num_objects{0};

static std::vector<std::unique_ptr<biosoup::NucleicAcid>> Load(const std::string& path) {
  const bool fq = path.find(".fastq") != std::string::npos || path.find(".fq") != std::string::npos;
  if (fq) {
    auto p = bioparser::Parser<biosoup::NucleicAcid>::Create<bioparser::FastqParser>(path);
    return p->Parse(-1);
  }
  auto p = bioparser::Parser<biosoup::NucleicAcid>::Create<bioparser::FastaParser>(path);
  return p->Parse(-1);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1];
  auto tp = std::make_shared<thread_pool::ThreadPool>(1);
  if (mode == "map" && argc >= 7) {
    auto reads = Load(argv[2]);
    const bool minhash = std::atoi(argv[6]) != 0;
    ram::MinimizerEngine me{tp, static_cast<std::uint32_t>(std::atoi(argv[3])), static_cast<std::uint32_t>(std::atoi(argv[4]))};
    me.Minimize(reads.begin(), reads.end(), minhash);
    me.Filter(std::atof(argv[5]));
    for (const auto& it : reads) {
      for (const auto& o : me.Map(it, true, true, minhash)) {
        std::printf("%u %u %u %u %u %u %u %u\n", o.lhs_id, o.lhs_begin, o.lhs_end, o.rhs_id, o.rhs_begin, o.rhs_end, o.score,
                    static_cast<unsigned>(o.strand));
      }
    }
    return 0;
  }
  if (mode == "edlib" && argc >= 4) {
    auto reads = Load(argv[2]);
    const std::size_t n = std::min<std::size_t>(std::atoi(argv[3]), reads.size() - 1);
    for (std::size_t i = 0; i < n; ++i) {
      const std::string a = reads[i]->InflateData(), b = reads[i + 1]->InflateData();
      EdlibAlignResult r = edlibAlign(a.c_str(), a.size(), b.c_str(), b.size(), edlibDefaultAlignConfig());
      std::printf("%d\n", r.editDistance);
      edlibFreeAlignResult(r);
    }
    return 0;
  }
  if (mode == "polish" && argc >= 4) {
    auto reads = Load(argv[2]);
    auto targets = Load(argv[3]);
    auto polisher = racon::Polisher::Create(tp, 0.0, 0.3, 500, true, 3, -5, -4);  // raven::Polish, RavenLib/src/polish.cc:43-51
    for (const auto& it : polisher->Polish(targets, reads, false)) std::printf("%s\n", it->InflateData().c_str());
    return 0;
  }
  return 2;
}
