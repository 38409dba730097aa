// edlib_dropin.hip — the edlib C entry points Raven's overlap path calls (include/edlib.h), on top of the batched
// device edit distance (rvn_edit_distance_batch).  Host code only: packing, a combining queue, result structs.
//
// edlibAlign is a blocking single-pair call that the reference issues from many pool threads at once
// (RavenLib/src/construct.cc:167-212, :374-429).  One launch per pair would be all latency, so concurrent calls are
// combined: the first caller to find no batch in flight becomes the leader, waits a few tens of microseconds for
// company, then runs everything queued as ONE upload + ONE kernel batch and wakes the others.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/edlib.h"
#include "../../include/raven_hip.h"

namespace {

struct Request {
  std::vector<uint64_t> q_words, t_words;
  uint32_t q_len = 0, t_len = 0;
  uint32_t distance = 0;
  int rc = RVN_OK;
  bool done = false;
};

struct Service {
  std::mutex mu;
  std::condition_variable cv_more, cv_done;
  std::vector<Request*> queue;
  bool leader_active = false;
  rvn_engine* engine = nullptr;
  bool engine_failed = false;

  ~Service() {
    // the engine is deliberately not destroyed: static destruction order vs the HIP runtime is undefined
  }

  int ensure_engine() {  // called by the leader only
    if (engine) return RVN_OK;
    if (engine_failed) return RVN_ENODEVICE;
    int device = 0;
    if (const char* ev = std::getenv("RVN_EDLIB_DEVICE")) device = std::atoi(ev);
    const int rc = rvn_engine_create(&engine, 15, 5, 500, 4, 100, 10000, device);
    if (rc != RVN_OK) {
      engine = nullptr;
      engine_failed = true;
      std::fprintf(stderr, "[raven_hip] edlibAlign: %s\n", rvn_last_error());
    }
    return rc;
  }

  void run(std::vector<Request*>& batch) {
    int rc = ensure_engine();
    if (rc == RVN_OK) {
      // one read set: query of request i = read 2i, target = read 2i+1
      std::vector<uint64_t> packed, woff(1, 0);
      std::vector<uint32_t> lens;
      std::vector<rvn_ed_pair> pairs(batch.size());
      for (size_t i = 0; i < batch.size(); ++i) {
        Request* r = batch[i];
        packed.insert(packed.end(), r->q_words.begin(), r->q_words.end());
        woff.push_back(packed.size());
        lens.push_back(r->q_len);
        packed.insert(packed.end(), r->t_words.begin(), r->t_words.end());
        woff.push_back(packed.size());
        lens.push_back(r->t_len);
        pairs[i] = rvn_ed_pair{static_cast<uint32_t>(2 * i), 0, r->q_len, static_cast<uint32_t>(2 * i + 1), 0, r->t_len, 1, 0};
      }
      packed.push_back(0);
      rvn_reads* reads = nullptr;
      rc = rvn_reads_upload(engine, packed.data(), packed.size() - 1, woff.data(), lens.data(), nullptr,
                            static_cast<uint32_t>(lens.size()), &reads);
      if (rc == RVN_OK) {
        std::vector<uint32_t> dist(batch.size());
        rc = rvn_edit_distance_batch(engine, reads, pairs.data(), static_cast<uint32_t>(pairs.size()), dist.data(),
                                     nullptr, nullptr);
        if (rc == RVN_OK)
          for (size_t i = 0; i < batch.size(); ++i) batch[i]->distance = dist[i];
      }
      rvn_reads_destroy(reads);
    }
    for (Request* r : batch) r->rc = rc;
  }

  void submit(Request* req) {
    std::unique_lock<std::mutex> lk(mu);
    queue.push_back(req);
    if (leader_active) {
      cv_more.notify_one();
      cv_done.wait(lk, [&] { return req->done; });
      return;
    }
    leader_active = true;
    // a short window for concurrent callers to join the first batch
    cv_more.wait_for(lk, std::chrono::microseconds(40), [&] { return queue.size() >= 512; });
    while (!queue.empty()) {
      std::vector<Request*> batch;
      batch.swap(queue);
      lk.unlock();
      // nothing may escape from here: a follower waits on `done`, and this is reached through extern "C" edlibAlign.
      // An exception in the host-side staging (std::bad_alloc, ...) fails the batch instead of stranding the followers.
      try {
        run(batch);
      } catch (const std::bad_alloc&) {
        for (Request* r : batch) r->rc = RVN_ENOMEM;
      } catch (...) {
        for (Request* r : batch) r->rc = RVN_EHIP;
      }
      lk.lock();
      for (Request* r : batch) r->done = true;
      cv_done.notify_all();
    }
    leader_active = false;
  }
};

Service& service() {
  static Service* s = new Service();  // never destroyed (see ~Service)
  return *s;
}

// bytes -> 2-bit codes by order of first appearance; false when more than 4 distinct symbols
bool pack2(const char* s, int n, int (&code_of)[256], int& n_symbols, std::vector<uint64_t>& words) {
  words.assign((static_cast<size_t>(n) + 31) / 32 + 1, 0);
  for (int i = 0; i < n; ++i) {
    const unsigned char c = static_cast<unsigned char>(s[i]);
    int code = code_of[c];
    if (code < 0) {
      if (n_symbols == 4) return false;
      code = code_of[c] = n_symbols++;
    }
    words[i >> 5] |= static_cast<uint64_t>(code) << ((i << 1) & 63);
  }
  words.pop_back();
  return true;
}

EdlibAlignResult error_result() {
  EdlibAlignResult r;
  std::memset(&r, 0, sizeof(r));
  r.status = EDLIB_STATUS_ERROR;
  r.editDistance = -1;
  return r;
}

}  // namespace

extern "C" {

EdlibAlignConfig edlibNewAlignConfig(int k, EdlibAlignMode mode, EdlibAlignTask task,
                                     const EdlibEqualityPair* additionalEqualities, int additionalEqualitiesLength) {
  EdlibAlignConfig c;
  c.k = k;
  c.mode = mode;
  c.task = task;
  c.additionalEqualities = additionalEqualities;
  c.additionalEqualitiesLength = additionalEqualitiesLength;
  return c;
}

EdlibAlignConfig edlibDefaultAlignConfig(void) {
  return edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, nullptr, 0);
}

void edlibFreeAlignResult(EdlibAlignResult result) {
  std::free(result.endLocations);
  std::free(result.startLocations);
  std::free(result.alignment);
}

EdlibAlignResult edlibAlign(const char* query, int queryLength, const char* target, int targetLength,
                            const EdlibAlignConfig config) {
  if (queryLength < 0 || targetLength < 0 || (queryLength && !query) || (targetLength && !target)) return error_result();
  if (config.mode != EDLIB_MODE_NW || config.task != EDLIB_TASK_DISTANCE || config.additionalEqualitiesLength != 0)
    return error_result();  // only the configuration Raven's hot path uses runs on the device; no CPU path here
  Request req;
  int code_of[256];
  for (int& c : code_of) c = -1;
  int n_symbols = 0;
  if (!pack2(query, queryLength, code_of, n_symbols, req.q_words) ||
      !pack2(target, targetLength, code_of, n_symbols, req.t_words))
    return error_result();
  req.q_len = static_cast<uint32_t>(queryLength);
  req.t_len = static_cast<uint32_t>(targetLength);
  service().submit(&req);
  if (req.rc != RVN_OK) return error_result();
  EdlibAlignResult r;
  std::memset(&r, 0, sizeof(r));
  r.status = EDLIB_STATUS_OK;
  r.alphabetLength = n_symbols;
  if (config.k >= 0 && req.distance > static_cast<uint32_t>(config.k)) {
    r.editDistance = -1;
    return r;
  }
  r.editDistance = static_cast<int>(req.distance);
  r.endLocations = static_cast<int*>(std::malloc(sizeof(int)));
  if (!r.endLocations) return error_result();
  r.endLocations[0] = targetLength - 1;
  r.numLocations = 1;
  return r;
}

char* edlibAlignmentToCigar(const unsigned char* alignment, int alignmentLength, EdlibCigarFormat cigarFormat) {
  if (cigarFormat != EDLIB_CIGAR_EXTENDED && cigarFormat != EDLIB_CIGAR_STANDARD) return nullptr;
  static const char ext[4] = {'=', 'I', 'D', 'X'}, stdc[4] = {'M', 'I', 'D', 'M'};
  const char* tab = cigarFormat == EDLIB_CIGAR_EXTENDED ? ext : stdc;
  std::vector<char> out;
  int run = 0;
  char last = 0;
  for (int i = 0; i <= alignmentLength; ++i) {
    char c = 0;
    if (i < alignmentLength) {
      if (alignment[i] > 3) return nullptr;
      c = tab[alignment[i]];
    }
    if (i == alignmentLength || (run && c != last)) {
      if (run) {
        char buf[16];
        const int n = std::snprintf(buf, sizeof(buf), "%d", run);
        out.insert(out.end(), buf, buf + n);
        out.push_back(last);
      }
      run = 0;
    }
    last = c;
    ++run;
  }
  char* s = static_cast<char*>(std::malloc(out.size() + 1));
  if (!s) return nullptr;
  std::memcpy(s, out.data(), out.size());
  s[out.size()] = 0;
  return s;
}

}  // extern "C"
