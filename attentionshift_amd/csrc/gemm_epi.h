// Epilogue arithmetic shared by the GEMM kernels (gemm.hip, gemm_pp.hip): the erf-GELU for a bf16 result, its derivative, and
// the two training-path forms applied to a 16-byte chunk.  ONE definition, so that every kernel that can serve a shape
// produces the same bits (reference: nn.GELU in Mlp, models/vision_transformer.py:47-59).
#pragma once
#include "common.h"

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-GELU for a bf16 result: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16's 2^-9), ~15
// instructions instead of the ~50 of erff -- at 64 outputs per lane erff alone cost twice the MFMA time of a K=768 tile.
__device__ __forceinline__ float gelu_bf16(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);   // erf(|x| / sqrt 2)
  return 0.5f * x + 0.5f * fabsf(x) * e;
}

// The same function on two values at once: the polynomial, the scalings and the final blend go through the packed fp32 VALU
// ops (v_pk_mul / v_pk_fma_f32: two IEEE results per instruction), the reciprocal and the exponential stay scalar (no packed
// transcendental) -- 17 instructions per PAIR instead of ~16 per value.  Used by the inference epilogue (act == 1), where a
// 256 x 256 tile spends 64 k of these per workgroup with the matrix pipe idle.
typedef __attribute__((ext_vector_type(2))) float g_f32x2;
__device__ __forceinline__ g_f32x2 gelu_bf16_x2(g_f32x2 x) {
  const g_f32x2 hx = x * 0.5f;
  g_f32x2 ahx;
  ahx.x = fabsf(hx.x); ahx.y = fabsf(hx.y);                             // |x| / 2
  const g_f32x2 z = ahx * 1.41421356237309504880f;                      // |x| / sqrt 2
  const g_f32x2 den = z * 0.3275911f + 1.0f;
  g_f32x2 t;
  t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
  g_f32x2 p = t * 1.061405429f + (-1.453152027f);
  p = p * t + 1.421413741f;
  p = p * t + (-0.284496736f);
  p = p * t + 0.254829592f;
  const g_f32x2 a = (z * z) * (-1.44269504088896340736f);
  g_f32x2 g;
  g.x = __builtin_amdgcn_exp2f(a.x); g.y = __builtin_amdgcn_exp2f(a.y);
  const g_f32x2 e = 1.0f - (p * t) * g;                                 // erf(|x| / sqrt 2)
  return hx + ahx * e;
}

// d/dx of the erf-GELU with the same erf: 0.5 (1 + erf(x / sqrt 2)) + x exp(-x^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_grad_bf16(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float g = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);                   // exp(-x^2 / 2)
  const float e = 1.0f - p * t * g;                                                            // erf(|x| / sqrt 2)
  return 0.5f + copysignf(0.5f * e, x) + x * g * 0.39894228040143267794f;
}
// the two training-path epilogues of the bf16 LDS-DMA kernel, applied to a staged 16-byte chunk (8 bf16) at copy-out:
//   act 2: the chunk is the PRE-activation h: write it to `pre` and gelu(h) to out     (fc1 under autograd)
//   act 3: the chunk is dA; out = dA * gelu'(h) with h read from `pre`                 (fc2's input gradient)
typedef __attribute__((ext_vector_type(2))) __bf16 g_bf16x2;
__device__ __forceinline__ uint4 gelu_chunk(uint4 hv) {
  const uint32_t w[4] = {hv.x, hv.y, hv.z, hv.w};
  uint32_t r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = gelu_bf16(__uint_as_float(w[j] << 16)), b = gelu_bf16(__uint_as_float(w[j] & 0xffff0000u));
    const g_bf16x2 pk = {(__bf16)a, (__bf16)b};
    r[j] = __builtin_bit_cast(uint32_t, pk);
  }
  return make_uint4(r[0], r[1], r[2], r[3]);
}
__device__ __forceinline__ uint4 dgelu_chunk(uint4 dv, uint4 hv) {
  const uint32_t d[4] = {dv.x, dv.y, dv.z, dv.w}, w[4] = {hv.x, hv.y, hv.z, hv.w};
  uint32_t r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = __uint_as_float(d[j] << 16) * gelu_grad_bf16(__uint_as_float(w[j] << 16));
    const float b = __uint_as_float(d[j] & 0xffff0000u) * gelu_grad_bf16(__uint_as_float(w[j] & 0xffff0000u));
    const g_bf16x2 pk = {(__bf16)a, (__bf16)b};
    r[j] = __builtin_bit_cast(uint32_t, pk);
  }
  return make_uint4(r[0], r[1], r[2], r[3]);
}

