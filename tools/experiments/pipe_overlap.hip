// Do the matrix pipe and the transcendental / plain VALU of ONE SIMD overlap?  (round 6: the SDPA forward's ablation costs ADD UP --
// MFMAs 36 %, softmax VALU 25 %, ... -- as if nothing ran beside anything.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pipe_overlap tools/experiments/pipe_overlap.hip && /tmp/pipe_overlap
// One workgroup of 8 waves per CU on every CU; wave w sits on SIMD w % 4, so waves 0-3 and 4-7 pair up on the four SIMDs.
// role[w]: 0 idle, 1 = ITER x 8 independent v_mfma_f32_32x32x16_bf16, 2 = ITER x 16 v_exp_f32, 3 = ITER x 24 v_fma_f32,
//          4 = both in ONE wave: per iteration 8 MFMAs with 16 exp + 24 fma between them (the SDPA step's mix).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void k(float* out, int iters, int roleA, int roleB) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB;
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
  float e[16], f[8];
  for (int i = 0; i < 16; ++i) e[i] = 0.001f * (threadIdx.x + i);
  for (int i = 0; i < 8; ++i) f[i] = 0.5f + 0.001f * i;
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], 1.0001f, 0.25f);
    }
  } else if (role == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        e[2 * i] = __builtin_amdgcn_exp2f(e[2 * i]);
        e[2 * i + 1] = __builtin_amdgcn_exp2f(e[2 * i + 1]);
        f[i] = fmaf(f[i], 1.0001f, 0.25f);
        f[(i + 3) & 7] = fmaf(f[(i + 3) & 7], 1.0001f, 0.25f);
        f[(i + 5) & 7] = fmaf(f[(i + 5) & 7], 1.0001f, 0.25f);
      }
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += e[i];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(float* out, int a, int b) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, a, b);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, a, b);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;   // ns per iteration
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const char* nm[] = {"idle", "8 mfma", "16 exp", "24 fma", "8 mfma + 16 exp + 24 fma in one wave"};
  const int cases[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {1, 3}, {2, 3}, {4, 0}, {4, 4}};
  for (auto& c : cases) {
    const float ns = run(out, c[0], c[1]);
    printf("{\"waves0_3\": \"%s\", \"waves4_7\": \"%s\", \"ns_per_iteration\": %.1f, \"cycles_at_2.4GHz\": %.0f}\n", nm[c[0]], nm[c[1]], ns, ns * 2.4f);
  }
  return 0;
}
