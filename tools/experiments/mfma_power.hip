// Does the MFMA instruction shape change what the chip sustains under its power / clock limit?
// Pure matrix-pipe loops on random bf16 operands (no memory traffic in the loop): v_mfma_f32_32x32x16_bf16 with 4
// independent accumulator chains per wave against v_mfma_f32_16x16x32_bf16 with 8 (the same 64 accumulator registers,
// the same flops per iteration), 1 / 2 / 4 waves per SIMD on all 256 CUs.  Prints TFLOP/s; run tools/experiments/
// power_probe.py-style rocm-smi sampling beside it for clock and power.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/experiments/mfma_power.hip && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256) void mfma_loop(const bf16x8* __restrict__ ops, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = ops[(t * 8 + i) & 65535]; b[i] = ops[(t * 8 + 4 + i) & 65535]; }
  float s = 0.0f;
  if (SHAPE == 0) {
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[i], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
  } else {
    f32x4 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + u) & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][3];
  }
  out[t] = s;
}

int main(int argc, char** argv) {
  const int secs_iters = argc > 1 ? atoi(argv[1]) : 20000;
  std::vector<unsigned short> h(65536 * 8);
  srand(1);
  for (auto& v : h) {                                      // random bf16 in about [-2, 2): sign, exponent 125..127, mantissa
    const unsigned sign = rand() & 1, exp = 125 + rand() % 3, man = rand() & 127;
    v = (unsigned short)((sign << 15) | (exp << 7) | man);
  }
  bf16x8* ops;
  float* out;
  hipMalloc(&ops, h.size() * 2);
  hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
  hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; wps *= 2) {                  // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    for (int shape = 0; shape < 2; ++shape) {
      const int grid = 256 * wps;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(grid), dim3(256), 0, 0, ops, out, secs_iters);
        else hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, ops, out, secs_iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // flops per wave per iteration: 16 x 32768 (32x32x16) = 32 x 16384 (16x16x32) = 524288
        const double fl = (double)grid * 4 * secs_iters * 524288.0;
        if (rep == 2)
          printf("{\"shape\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.2f, \"tflops\": %.0f}\n", shape == 0 ? "32x32x16" : "16x16x32", wps, ms,
                 fl / ms / 1e9);
      }
    }
  }
  return 0;
}
