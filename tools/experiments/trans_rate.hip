// Issue rate of v_exp_f32 against plain VALU on gfx950, and whether they overlap inside one wave / across the waves of a SIMD:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/trans_rate tools/experiments/trans_rate.hip && /tmp/trans_rate
// Each kernel runs ITER iterations of 16 independent chains per lane; one workgroup of W waves per CU on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.001f * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);                        // 16 exp
      if (MODE == 1) a[i] = fmaf(a[i], 1.0001f, 0.5f);                           // 16 fma
      if (MODE == 2) { if (i & 1) a[i] = __builtin_amdgcn_exp2f(a[i]); else a[i] = fmaf(a[i], 1.0001f, 0.5f); }   // 8 + 8
      if (MODE == 3) { a[i] = __builtin_amdgcn_exp2f(a[i]); a[i] = fmaf(a[i], 1.0001f, 0.5f); }                   // 16 + 16 (dependent pairs, 16 chains)
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int waves, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ops = (MODE == 3 ? 32.0 : 16.0) * iters;              // wave-instructions per wave
  // cycles per wave-instruction per SIMD at 2.4 GHz nominal: waves per SIMD = waves / 4
  const double cyc = ms * 1e-3 * 2.4e9 / (ops * (waves / 4.0));
  printf("{\"kernel\": \"%s\", \"waves_per_cu\": %d, \"ms\": %.3f, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.2f}\n", name, waves, ms, cyc);
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  for (int waves : {4, 8, 16}) {
    run<0>("16 v_exp_f32", waves, out);
    run<1>("16 v_fma_f32", waves, out);
    run<2>("8 exp + 8 fma interleaved", waves, out);
    run<3>("16 exp + 16 fma (dependent pairs)", waves, out);
  }
  return 0;
}
