// pmc_calibrate.hip — what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of this repository
// (the guide's "x 2" for FETCH_SIZE is calibrated on wide coalesced streams only).  Five kernels over a
// table far beyond the Infinity Cache, each with a known byte count; tools/pmc_calibrate.sh runs the binary under two
// --pmc passes and prints counter bytes / known bytes per pattern.  Measurement tool, not part of the library.
//   stream_read16   every lane reads 16 B, consecutive                      known = table bytes
//   gather4         every lane reads 4 B at a random word                   known = probes x 4 B (x 32 / 64 / 128: sector, line)
//   gather16        every lane reads 16 B at a random 16-B slot             known = probes x 16 B
//   stream_write16  every lane writes 16 B, consecutive                     known = table bytes
//   scatter2        every lane writes 2 B at a random halfword              known = probes x 2 B
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
__global__ void stream_read16(const uint4* __restrict__ t, size_t n16, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const uint4 v = t[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void gather4(const unsigned* __restrict__ t, size_t n4, size_t probes, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < probes; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    acc += t[mix(i) % n4];
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void gather16(const uint4* __restrict__ t, size_t n16, size_t probes, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < probes; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const uint4 v = t[mix(i) % n16];
    acc += v.x ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void stream_write16(uint4* __restrict__ t, size_t n16) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    t[i] = uint4{static_cast<unsigned>(i), 1u, 2u, 3u};
}
__global__ void scatter2(unsigned short* __restrict__ t, size_t n2, size_t probes) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < probes; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    t[mix(i) % n2] = static_cast<unsigned short>(i);
}
int main() {
  const size_t bytes = 8ULL << 30, probes = 1ULL << 28;
  void* t = nullptr;
  unsigned* sink = nullptr;
  CK(hipMalloc(&t, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(t, 1, bytes));
  const int blocks = 256 * 16, threads = 256;
  for (int rep = 0; rep < 2; ++rep) {
    stream_read16<<<blocks, threads>>>(static_cast<const uint4*>(t), bytes / 16, sink);
    gather4<<<blocks, threads>>>(static_cast<const unsigned*>(t), bytes / 4, probes, sink);
    gather16<<<blocks, threads>>>(static_cast<const uint4*>(t), bytes / 16, probes, sink);
    stream_write16<<<blocks, threads>>>(static_cast<uint4*>(t), bytes / 16);
    scatter2<<<blocks, threads>>>(static_cast<unsigned short*>(t), bytes / 2, probes);
  }
  CK(hipDeviceSynchronize());
  std::printf("{\"table_bytes\": %zu, \"probes\": %zu}\n", bytes, probes);
  return 0;
}
