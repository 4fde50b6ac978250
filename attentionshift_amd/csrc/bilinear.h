// ATen-exact bilinear interpolation (align_corners = False) shared by ccl.hip and refine.hip.
// Both are compiled with -ffp-contract=off: src = scale*(dst+0.5)-0.5 must round as separate mul/sub,
// and the two interpolation FMAs are explicit (see oracle upsample_bilinear_explicit, which tests pin
// bit-for-bit to F.interpolate on CPU).
#pragma once
#include "common.h"

struct Lerp { int i0, i1; float l0, l1; };

__device__ __forceinline__ Lerp lerp_axis(int dst, int n_in, float scale) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = fmaxf(src, 0.0f);
  Lerp r;
  r.i0 = min((int)src, n_in - 1);
  r.i1 = min(r.i0 + 1, n_in - 1);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.0f - r.l1;
  return r;
}
__device__ __forceinline__ float bilerp(const float* __restrict__ m, int Wp, const Lerp& ly, const Lerp& lx) {
  const float top = fmaf(lx.l0, m[ly.i0 * Wp + lx.i0], lx.l1 * m[ly.i0 * Wp + lx.i1]);
  const float bot = fmaf(lx.l0, m[ly.i1 * Wp + lx.i0], lx.l1 * m[ly.i1 * Wp + lx.i1]);
  return fmaf(ly.l0, top, ly.l1 * bot);
}

// order-preserving float <-> uint map so min/max can use integer atomics (deterministic)
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Row-block evaluation of the x`up` bilinear map.  bilerp(y, x) = fma(ly.l0, top, ly.l1 * bot) where
// top / bot = the horizontal interpolation in source rows ly.i0 / ly.i1 depend on (x, source row) only: a thread
// keeps the columns it owns, walks the rows of its block and recomputes top / bot only when the source row pair
// changes (every `up` rows) -- 2 flops per pixel instead of two index computations, four loads and 7 flops, with
// exactly the same operations and rounding as bilerp().
struct ColLerp { Lerp lx; float top, bot; };
__device__ __forceinline__ void col_refresh(ColLerp& c, const float* __restrict__ src, int Wp, const Lerp& ly) {
  c.top = fmaf(c.lx.l0, src[ly.i0 * Wp + c.lx.i0], c.lx.l1 * src[ly.i0 * Wp + c.lx.i1]);
  c.bot = fmaf(c.lx.l0, src[ly.i1 * Wp + c.lx.i0], c.lx.l1 * src[ly.i1 * Wp + c.lx.i1]);
}
__device__ __forceinline__ float col_value(const ColLerp& c, const Lerp& ly) { return fmaf(ly.l0, c.top, ly.l1 * c.bot); }

