// Dependent-chain floor of a latency-bound launch sequence (the shape of as_cosine_shift: 16 dependent launches of small
// grids whose threads walk kernel arguments -> box -> operands -> result).  Each launch of `hop_kernel<H>` makes H
// DEPENDENT global loads per thread (the address of load i+1 comes from load i; the buffers are L2-resident, a few KB)
// and one store that the next launch's first load reads, so nothing overlaps across launches.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_floor tools/experiments/chain_floor.hip && /tmp/chain_floor
//
// Prints one JSON line: microseconds per chain of 16 launches for H = 0 .. 4 hops, grid = 126 workgroups of 256 threads
// (21 tiles x 6 objects, config 2), median of 200 chains; every chain is queued behind a ~300 us blocker kernel so that
// the host's launch rate does not enter.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int H>
__global__ void hop_kernel(const int* __restrict__ idx, const float* __restrict__ prev, float* __restrict__ out, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  int i = (t + (int)prev[t % n]) % n;                   // first hop depends on the previous launch's result
  float v = 0.0f;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    i = idx[i];                                          // dependent: the address comes from the previous load
    v += (float)(i & 1);
  }
  out[t % n] = v;                                        // 0 .. H: the next launch's first index moves by a few slots
}

// keeps the stream busy while the host queues a whole chain behind it, so the chain then runs at the DEVICE's pace (an
// eager host loop tops out at ~2.8 us per launch, which would hide the device-side boundary)
__global__ void blocker_kernel(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink != nullptr && threadIdx.x == 12345) sink[0] = 1;
}

template <int H> float chain_us(int launches, int grid, int* idx, float* a, float* b, int n, hipStream_t s) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> t;
  for (int rep = 0; rep < 220; ++rep) {
    hipLaunchKernelGGL(blocker_kernel, dim3(1), dim3(64), 0, s, (long long)30000, (int*)nullptr);   // ~300 us at 100 MHz
    hipEventRecord(e0, s);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(hop_kernel<H>, dim3(grid), dim3(256), 0, s, idx, (l & 1) ? b : a, (l & 1) ? a : b, n);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 20) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const int n = 1 << 15, grid = 126, launches = 16;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 13) % n);
  int* idx;
  float *a, *b;
  hipMalloc(&idx, n * sizeof(int));
  hipMalloc(&a, n * sizeof(float));
  hipMalloc(&b, n * sizeof(float));
  hipMemcpy(idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice);
  hipMemset(a, 0, n * sizeof(float));
  hipMemset(b, 0, n * sizeof(float));
  hipStream_t s;
  hipStreamCreate(&s);
  const float t0 = chain_us<0>(launches, grid, idx, a, b, n, s), t1 = chain_us<1>(launches, grid, idx, a, b, n, s),
              t2 = chain_us<2>(launches, grid, idx, a, b, n, s), t3 = chain_us<3>(launches, grid, idx, a, b, n, s),
              t4 = chain_us<4>(launches, grid, idx, a, b, n, s);
  printf("{\"launches\": %d, \"grid\": %d, \"us_per_chain_by_hops\": {\"0\": %.1f, \"1\": %.1f, \"2\": %.1f, \"3\": %.1f, \"4\": %.1f}, "
         "\"us_per_launch_boundary\": %.2f, \"us_per_dependent_hop\": %.2f}\n",
         launches, grid, t0, t1, t2, t3, t4, t0 / launches, (t4 - t0) / (4.0 * launches));
  return 0;
}
