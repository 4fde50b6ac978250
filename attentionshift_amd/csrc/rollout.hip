// Row-sliced attention roll-out for gfx950.
//
// The reference keeps attn.mean(1) ([B,N,N]) of every layer (visual_transformer_det.py:236,242) and
// multiplies Lc of them as dense N^3 GEMMs (stdroi_point_deform_attn_reppoints.py:1257-1272) only to
// read rows [-T:] of every partial product (stdroi:2272).  Row slicing commutes with the chain, so
// here a [T,N] matrix R per image is pushed through the layers top-down:
//     A_hat = (mean_h P + I) / rowsum,  rowsum == 2  =>  R_out = 0.5 * (R_in . mean_h P + R_in)
// and the tiles of mean_h P are RECOMPUTED from q, k and the saved log-sum-exp (never stored):
//     P_h[i][j] = exp(q_i.k_j/8 - lse_h[i]).
// pbar_tile(): one wave, one 32x32 tile of mean_h P via h x 4 MFMAs + exp2, accumulators laid out
// rows = contraction index (registers), cols = output column (lanes) so they feed the second MFMA
// as its B operand without any data movement (same trick as sdpa.hip).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int HD = 64, RO_NT = 256;
constexpr float LOG2E = 1.44269504088896340736f;

template <typename T> struct KjCfg { static constexpr int PITCH = HD * (int)sizeof(T) + 16; };

// Fragment-major copy of a roll-out matrix R[b][i][k] (rows i < 128, contraction index k): the 8 values lane
// (i%32, half) needs as the A operand of k16 sub-step s2 of block (kb = k/32, ib = i/32) are contiguous and a wave
// reads 1 KiB contiguous: [b][kb][ib][s2][lane][8].  Element (i, k) -> its slot:
__device__ __forceinline__ size_t rf_slot(int b, int nkb, int i, int k) {
  const int rem = k & 31, s2 = rem >> 4, r16 = rem & 15;
  const int lane = (i & 31) + 32 * ((r16 & 7) >> 2), t = (r16 >> 3) * 4 + (r16 & 3);
  return ((((((size_t)b * nkb + (k >> 5)) * 4 + (i >> 5)) * 2 + s2) * 64) + lane) * 8 + t;
}
__device__ __forceinline__ size_t rf_frag(int b, int nkb, int kb, int ib, int s2, int lane) {
  return ((((((size_t)b * nkb + kb) * 4 + ib) * 2 + s2) * 64) + lane) * 8;
}

// mean over heads of the softmax tile rows [i0, i0+32) x cols [j0, j0+32) for image b.
// kj_lds: this workgroup's K rows j0..j0+31 of every head, [h][32][PITCH]  (or nullptr: read global)
template <typename T>
__device__ __forceinline__ f32x16 pbar_tile(const T* __restrict__ q, const T* __restrict__ k,
                                            const float* __restrict__ lse, const char* kj_lds, int b, int h,
                                            int N, int Npad, int i0, int j0, int li, int half) {
  f32x16 pbar;
#pragma unroll
  for (int r = 0; r < 16; ++r) pbar[r] = 0.0f;
  const int irow = min(i0 + li, N - 1);
  const int jrow = min(j0 + li, N - 1);
  for (int hh = 0; hh < h; ++hh) {
    const size_t bh = (size_t)b * h + hh;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag<T> fa, fb;
      fa.load16B(q + qf_frag(bh, Npad, irow, ks, half));
      if (kj_lds != nullptr)
        fb.load16B(reinterpret_cast<const T*>(kj_lds + ((size_t)hh * 32 + li) * KjCfg<T>::PITCH) + ks * 16 + half * 8);
      else
        fb.load16B(k + (bh * Npad + jrow) * HD + ks * 16 + half * 8);
      s = mma32(fa, fb, s);
    }
    const float* lrow = lse + bh * N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(i0 + acc_row(r, half), N - 1);
      pbar[r] += __builtin_amdgcn_exp2f(s[r] - lrow[row] * LOG2E);          // q is pre-scaled: s = log2(e) q.k / 8
    }
  }
  const float inv_h = 1.0f / (float)h;
#pragma unroll
  for (int r = 0; r < 16; ++r) pbar[r] *= inv_h;
  return pbar;
}

// out[b, i, j] = mean_h P[row0 + i][j]            (TOP: 0.5 * (that + [row0 + i == j]))
template <typename T, bool TOP>
__global__ __launch_bounds__(RO_NT) void attn_mean_rows_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const float* __restrict__ lse,
                                                               float* __restrict__ out, T* __restrict__ rf_out, int B,
                                                               int N, int Npad, int h, int row0, int nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int jb = blockIdx.x * 4 + wave, ib = blockIdx.y, b = blockIdx.z;
  const int j0 = jb * 32, i0 = ib * 32;
  if (j0 >= N) return;
  const f32x16 p = pbar_tile<T>(q, k, lse, nullptr, b, h, N, Npad, row0 + i0, j0, li, half);
  const int j = j0 + li;
  const int nkb = (N + 31) / 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + acc_row(r, half);
    float v = p[r];
    if (TOP) v = 0.5f * (v + ((row0 + i) == j ? 1.0f : 0.0f));
    const bool live = i < nrows && j < N;
    if (live) out[((size_t)b * nrows + i) * N + j] = v;
    if (rf_out != nullptr && i < 128) rf_out[rf_slot(b, nkb, i, j)] = from_f32<T>(live ? v : 0.0f);
  }
}

// ---------------------------------------------------------------------------------------------------------
// rollout_step2: latency-tolerant version.  All four waves of a workgroup work on the SAME 32-row contraction
// block at a time, each on its own heads (wave w: heads w, w+4, ...):
//   * the per-row -lse term is injected with ONE exact-fp32 MFMA per head (A = -log2(e)*lse[row] in the k=0 slot,
//     B = 1), which lands it directly in the accumulator layout -- no per-register lse loads;
//   * Q fragments of the next block are prefetched into registers while the current block is computed (bf16);
//   * the four partial head sums are exchanged through a double-buffered 16 KiB LDS slab (one barrier per
//     block) and every wave then owns ONE 32-row block of R for the second MFMA, so no final reduction.
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct Kj2 {           // bf16: unpadded 128-B rows, XOR-swizzled 16-B chunks; f32: padded
  static constexpr int PITCH = sizeof(T) == 2 ? 128 : (HD * 4 + 16);
  __device__ static __forceinline__ int off(int row, int elem) {      // elem multiple of 8
    if (sizeof(T) == 2) return row * 128 + ((((elem >> 3) ^ (row & 7))) << 4);
    return row * PITCH + elem * 4;
  }
};

template <typename T, int HPW>
__global__ __launch_bounds__(RO_NT) void rollout_step2_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ Rin, const T* __restrict__ rf_in,
                                                              float* __restrict__ Rout, T* __restrict__ rf_out,
                                                              float* __restrict__ part, int B, int N, int Npad, int h,
                                                              int Trows, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool PREFETCH = sizeof(T) == 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int j0 = blockIdx.x * 32, b = blockIdx.y, split = blockIdx.z;
  char* kj = smem;
  float4* xchg = reinterpret_cast<float4*>(smem + (size_t)h * 32 * Kj2<T>::PITCH);    // [2][4 waves][4][64]

  {  // stage K rows j0..j0+31 of every head
    constexpr int CPR = HD * (int)sizeof(T) / 16;
    const int total = h * 32 * CPR;
    for (int c = tid; c < total; c += RO_NT) {
      const int hh = c / (32 * CPR), rem = c % (32 * CPR);
      const int jr = rem / CPR, ch = rem % CPR;
      const int jrow = min(j0 + jr, N - 1);
      const uint4 u = *reinterpret_cast<const uint4*>(
          reinterpret_cast<const char*>(k + (((size_t)b * h + hh) * Npad + jrow) * HD) + ch * 16);
      const int elem = ch * (16 / (int)sizeof(T));
      const int base = Kj2<T>::off(hh * 32 + jr, elem & ~7);
      *reinterpret_cast<uint4*>(kj + base + (elem & 7) * (int)sizeof(T)) = u;
    }
  }
  __syncthreads();

  const int nkb = (N + 31) / 32;
  const float inv_h = 1.0f / (float)h;
  const bool own_rows = wave * 32 < Trows;              // this wave's 32-row block of R exists

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  // everything a block needs from global memory (Q fragments, the lane's lse values, its R values) is fetched one
  // block AHEAD, in one batch at the top of the iteration: vmcnt completes in order, so a load issued at its point
  // of use would first have to drain the whole prefetch batch issued before it.
  struct Fetch {
    Frag<T> fq[HPW][4];
    float l8[HPW];
    Frag<T> fr[2];
  };
  auto fetch = [&](int kb, Fetch& f) {
    const int row = min(kb * 32 + li, N - 1);
#pragma unroll
    for (int t = 0; t < HPW; ++t) {
      const int hh = min(wave + 4 * t, h - 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) f.fq[t][ks].load16B(q + qf_frag((size_t)b * h + hh, Npad, row, ks, half));
      f.l8[t] = -LOG2E * lse[((size_t)b * h + hh) * N + row];
    }
    if (own_rows) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) f.fr[s2].load16B(rf_in + rf_frag(b, nkb, kb, wave, s2, lane));
    }
  };
  // contraction-range split: workgroup `split` of `nsplit` takes blocks [kb0, kb1) and leaves a raw partial sum; the
  // grid is then (N/32) x B x nsplit workgroups instead of 264 for 256 CUs (which ran as two rounds of one per CU)
  const int kb0 = (int)((long long)nkb * split / nsplit), kb1 = (int)((long long)nkb * (split + 1) / nsplit);
  Fetch cur, nxt;
  if (PREFETCH && kb0 < kb1) fetch(kb0, nxt);

  for (int kb = kb0; kb < kb1; ++kb) {
    const int k0 = kb * 32;
    if (PREFETCH) {
      cur = nxt;
      if (kb + 1 < kb1) fetch(kb + 1, nxt);
    } else {
      fetch(kb, cur);
    }

    f32x16 pbar;
#pragma unroll
    for (int r = 0; r < 16; ++r) pbar[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < HPW; ++t) {
      const int hh = wave + 4 * t;
      if (hh < h) {                                      // wave-uniform
        const float l8 = cur.l8[t];
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
        // sc[i][j] = -8 * lse[k0 + i]: exact fp32 row broadcast in accumulator layout
        sc = __builtin_amdgcn_mfma_f32_32x32x2f32(half == 0 ? l8 : 0.0f, 1.0f, sc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          Frag<T> fk;
          fk.load16B(reinterpret_cast<const T*>(kj + Kj2<T>::off(hh * 32 + li, ks * 16 + half * 8)));
          sc = mma32(cur.fq[t][ks], fk, sc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) pbar[r] += __builtin_amdgcn_exp2f(sc[r]);
      }
    }
    // publish this wave's partial head sum
    float4* slab = xchg + (size_t)((kb - kb0) & 1) * 4 * 4 * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      slab[(wave * 4 + g) * 64 + lane] = make_float4(pbar[4 * g], pbar[4 * g + 1], pbar[4 * g + 2], pbar[4 * g + 3]);
    __syncthreads();
    if (own_rows) {
      Frag<T> fp[2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = slab[(0 * 4 + g) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float4 u = slab[(w * 4 + g) * 64 + lane];
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int r = 4 * g + x;
          const bool live = k0 + acc_row(r, half) < N;   // contraction rows beyond N contribute nothing
          fp[r >> 3].set(r & 7, live ? e[x] * inv_h : 0.0f);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acc = mma32(cur.fr[s2], fp[s2], acc);
    }
  }

  if (own_rows) {
    const int j = j0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = wave * 32 + acc_row(r, half);
      const bool live = i < Trows && j < N;
      if (nsplit > 1) {
        if (live) part[(((size_t)split * B + b) * Trows + i) * N + j] = acc[r];
        continue;
      }
      float v = 0.0f;
      if (live) {
        const size_t idx = ((size_t)b * Trows + i) * N + j;
        v = 0.5f * (acc[r] + Rin[idx]);
        Rout[idx] = v;
      }
      if (rf_out != nullptr) rf_out[rf_slot(b, nkb, i, j)] = from_f32<T>(v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// rollout_step3: barrier-free inner loop.  Each WAVE takes whole 32-row contraction blocks (all heads) of the
// workgroup's range, so the head-mean tile never leaves its registers: no per-block LDS exchange, no per-block
// barrier, LDS holds only the K_j tile (48 KiB at h = 12).  Per block a wave runs, per head, one MFMA that injects
// -8*lse in accumulator layout (inject_rows) + 4 MFMAs q.k^T and 16 exp2, then 8 MFMAs R . Pbar for all four 32-row blocks
// of R.  Q fragments are prefetched one head ahead, R fragments at the top of the block.  The four waves'
// accumulators are summed once at the end (two passes through LDS, fixed order).
// ---------------------------------------------------------------------------------------------------------
#ifndef AS_ROLLOUT_ABLATE
#define AS_ROLLOUT_ABLATE 0                    // timing ablations (tools/experiments/rollout_ablate.py); wrong results when != 0
#endif
// -8*lse[row] enters the score accumulators through an MFMA (it has to land in accumulator layout: 16 different rows per
// lane).  fp32 tensors: one exact v_mfma_f32_32x32x2_f32.  bf16 tensors: the value is split into three bf16 terms
// (8 + 8 + 8 mantissa bits: hi + mid + lo == x to the last fp32 bit) on k-slots 0..2 of ONE bf16 MFMA against ones --
// half the matrix-pipe time of the fp32 instruction.
template <typename T> __device__ __forceinline__ f32x16 inject_rows(float x, int half);
template <> __device__ __forceinline__ f32x16 inject_rows<float>(float x, int half) {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.0f;
  return __builtin_amdgcn_mfma_f32_32x32x2f32(half == 0 ? x : 0.0f, 1.0f, z, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 inject_rows<__bf16>(float x, int half) {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.0f;
  x = half == 0 ? x : 0.0f;
  const __bf16 hi = (__bf16)x;
  const float r1 = x - (float)hi;
  const __bf16 mid = (__bf16)r1;
  const __bf16 lo = (__bf16)(r1 - (float)mid);
  const __bf16 zero = (__bf16)0.0f, one = (__bf16)1.0f;
  Frag<__bf16> fa, fb;
  fa.v = bf16x8{hi, mid, lo, zero, zero, zero, zero, zero};
  fb.v = bf16x8{one, one, one, zero, zero, zero, zero, zero};
  return mma32(fa, fb, z);
}

template <typename T>
__global__ __launch_bounds__(RO_NT, 3) void rollout_step3_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                 const float* __restrict__ lse,
                                                                 const float* __restrict__ Rin, const T* __restrict__ rf_in,
                                                                 float* __restrict__ Rout, T* __restrict__ rf_out,
                                                                 float* __restrict__ part, int B, int N, int Npad, int h,
                                                                 int Trows, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int j0 = blockIdx.x * 32, b = blockIdx.y, split = blockIdx.z;
  char* kj = smem;
  {  // stage K rows j0..j0+31 of every head
    constexpr int CPR = HD * (int)sizeof(T) / 16;
    const int total = h * 32 * CPR;
    for (int c = tid; c < total; c += RO_NT) {
      const int hh = c / (32 * CPR), rem = c % (32 * CPR);
      const int jr = rem / CPR, ch = rem % CPR;
      const int jrow = min(j0 + jr, N - 1);
      const uint4 u = *reinterpret_cast<const uint4*>(
          reinterpret_cast<const char*>(k + (((size_t)b * h + hh) * Npad + jrow) * HD) + ch * 16);
      const int elem = ch * (16 / (int)sizeof(T));
      const int base = Kj2<T>::off(hh * 32 + jr, elem & ~7);
      *reinterpret_cast<uint4*>(kj + base + (elem & 7) * (int)sizeof(T)) = u;
    }
  }
  __syncthreads();

  const int nkb = (N + 31) / 32;
  const int nib = (Trows + 31) / 32;                      // 32-row blocks of R that exist (<= 4)
  const float inv_h = 1.0f / (float)h;
  const int kb0 = (int)((long long)nkb * split / nsplit), kb1 = (int)((long long)nkb * (split + 1) / nsplit);

  f32x16 acc[4];
#pragma unroll
  for (int ib = 0; ib < 4; ++ib)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ib][r] = 0.0f;

  for (int kb = kb0 + wave; kb < kb1; kb += 4) {
    const int k0 = kb * 32;
    const int row = min(k0 + li, N - 1);
    Frag<T> fq[4];
    float l8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fq[ks].load16B(q + qf_frag((size_t)b * h, Npad, row, ks, half));
    l8 = -LOG2E * lse[((size_t)b * h) * N + row];
    f32x16 pbar;
#pragma unroll
    for (int r = 0; r < 16; ++r) pbar[r] = 0.0f;
    auto head = [&](int hh) {
      f32x16 sc = inject_rows<T>(AS_ROLLOUT_ABLATE == 3 ? 0.0f : l8, half);       // -log2(e)*lse[row i]
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<T> fk;
        fk.load16B(reinterpret_cast<const T*>(kj + Kj2<T>::off(hh * 32 + li, ks * 16 + half * 8)));
        if (AS_ROLLOUT_ABLATE != 4) sc = mma32(fq[ks], fk, sc);
        else sc[ks] += (float)fk.v[0];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) pbar[r] += AS_ROLLOUT_ABLATE == 2 ? sc[r] : __builtin_amdgcn_exp2f(sc[r]);
    };
    for (int hh = 0; hh + 1 < h; ++hh) {                  // heads 0 .. h-2: next head's Q fragments in flight
      Frag<T> fn[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (AS_ROLLOUT_ABLATE == 1) fn[ks] = fq[ks];
        else fn[ks].load16B(q + qf_frag((size_t)b * h + hh + 1, Npad, row, ks, half));
      }
      const float l8n = AS_ROLLOUT_ABLATE == 1 ? l8 : -LOG2E * lse[((size_t)b * h + hh + 1) * N + row];
      head(hh);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fq[ks] = fn[ks];
      l8 = l8n;
    }
    Frag<T> fr[4][2];                                      // last head: the R fragments take the prefetch slot
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      const int ibc = min(ib, nib - 1);                    // a missing block re-reads a valid one; its MFMA is skipped
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) fr[ib][s2].load16B(rf_in + rf_frag(b, nkb, kb, ibc, s2, lane));
    }
    head(h - 1);
    Frag<T> fp[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool live = k0 + acc_row(r, half) < N;         // contraction rows beyond N contribute nothing
      fp[r >> 3].set(r & 7, live ? pbar[r] * inv_h : 0.0f);
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
      if (ib < nib) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) acc[ib] = mma32(fr[ib][s2], fp[s2], acc[ib]);
      }
  }

  // sum the four waves' accumulators: two passes of two R blocks through LDS (32 KiB), wave w finishes block w
  float* red = reinterpret_cast<float*>(smem);
  f32x16 mine;
#pragma unroll
  for (int r = 0; r < 16; ++r) mine[r] = 0.0f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();                                      // K tile / previous pass no longer needed
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * 2 + x) * 16 + r) * 64 + lane] = acc[pass * 2 + x][r];
    __syncthreads();
    if ((wave >> 1) == pass) {
      const int x = wave & 1;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        mine[r] = ((red[((0 * 2 + x) * 16 + r) * 64 + lane] + red[((1 * 2 + x) * 16 + r) * 64 + lane]) +
                   red[((2 * 2 + x) * 16 + r) * 64 + lane]) + red[((3 * 2 + x) * 16 + r) * 64 + lane];
    }
  }
  if (wave * 32 < Trows) {
    const int j = j0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = wave * 32 + acc_row(r, half);
      const bool live = i < Trows && j < N;
      if (nsplit > 1) {
        if (live) part[(((size_t)split * B + b) * Trows + i) * N + j] = mine[r];
        continue;
      }
      float v = 0.0f;
      if (live) {
        const size_t idx = ((size_t)b * Trows + i) * N + j;
        v = 0.5f * (mine[r] + Rin[idx]);
        Rout[idx] = v;
      }
      if (rf_out != nullptr) rf_out[rf_slot(b, nkb, i, j)] = from_f32<T>(v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// rollout_step4 (bf16, h % 4 == 0): the step as a streamed-operand kernel.  rollout_step3 is bound by the L2 -> CU path:
// every 32-column workgroup re-reads ALL of Q (1.7 GB per launch at B = 2, h = 12, N = 4197;
// tools/experiments/rollout_ablate.py).  Here a workgroup owns 128 key columns x 4 HEADS: each of its four waves keeps
// the K fragments of ITS 32 columns in registers (4 heads x 4 k16 steps), and the 32-row contraction blocks -- the Q
// fragments of the four heads (16 KiB) and the R fragments (8 KiB), both already fragment-major in HBM -- stream once
// per workgroup through a three-stage LDS-DMA ring shared by the four waves (counted vmcnt + one raw barrier per block,
// as gemm.hip): a quarter of the bytes per MFMA.  The head groups and the contraction splits produce partial products
// (linear in the head mean: each carries its 1/h) that rollout_finish_kernel adds in a fixed order; a wave finishes its
// own 32 columns, so there is no cross-wave reduction.
// ---------------------------------------------------------------------------------------------------------
constexpr int R4_HPG = 4;                                  // heads per workgroup
constexpr int R4_QBYTES = R4_HPG * 4 * 1024;               // Q fragments of one contraction block: [head][k16 step][1 KiB]
constexpr int R4_LSE = R4_QBYTES + 8 * 1024;               // + R fragments [ib][s2][1 KiB]
constexpr int R4_STAGE = R4_LSE + 1024;                    // + lse rows [head][64 lanes] fp32 (lane l holds row l % 32)
// ring depth / waves per SIMD asked of hipcc, per form: the 32-row form (NIB = 1, 168 registers) runs three workgroups per
// CU on a two-stage ring (51 KB each); the 128-row form (NIB = 4) carries 64 more accumulator registers and keeps the
// three-stage ring with two workgroups per CU (at three it spills 53 registers: 1.89 vs 0.94 ms for the 7-layer roll-out)
#ifndef AS_R4_OCC1
#define AS_R4_OCC1 3                          // (experiments: waves per SIMD asked for the 32-row form)
#endif
template <int NIB> struct R4Cfg { static constexpr int NSTAGE = NIB == 1 ? 2 : 3, OCC = NIB == 1 ? AS_R4_OCC1 : 2; };

typedef __attribute__((ext_vector_type(4))) unsigned r4_u32x4;
template <int OFF> __device__ __forceinline__ void r4_lds_read128(r4_u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void r4_lds_read128_dyn(r4_u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
__device__ __forceinline__ void r4_wait_lds(r4_u32x4& a, r4_u32x4& b, r4_u32x4& c, r4_u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void r4_wait_lds8(r4_u32x4& a, r4_u32x4& b, r4_u32x4& c, r4_u32x4& d, r4_u32x4& e, r4_u32x4& f,
                                             r4_u32x4& g, r4_u32x4& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void r4_lds_read32(float& dst, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr));
}
__device__ __forceinline__ void r4_wait_lds2(r4_u32x4& a, r4_u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
  __builtin_amdgcn_sched_barrier(0);
}

// global_load_lds, saddr form: 64 lanes x 16 (4) B from sbase + voff[lane] -> LDS [dst + 16 (4) * lane]; m0 is saved and
// restored in the same statement; `sbase` and `dst` are wave-uniform (as sdpa.hip's lds_dma16)
__device__ __forceinline__ void r4_dma16(unsigned voff, const char* sbase, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
__device__ __forceinline__ void r4_dma4(unsigned voff, const char* sbase, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

template <int NIB>                                       // 32-row blocks of R carried (1: Trows <= 32, else 4)
__global__ __launch_bounds__(RO_NT, R4Cfg<NIB>::OCC) void rollout_step4_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                                 const float* __restrict__ lse, const __bf16* __restrict__ rf_in,
                                                                 float* __restrict__ part, int B, int N, int Npad, int h,
                                                                 int Trows, int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R4_NSTAGE = R4Cfg<NIB>::NSTAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int ngroups = h / R4_HPG;
  const int j0 = blockIdx.x * 128 + wave * 32, b = blockIdx.y;
  const int hg = blockIdx.z % ngroups, split = blockIdx.z / ngroups;
  const int nkb = (N + 31) / 32;
  const int nib = min((Trows + 31) / 32, NIB);
  const int kb0 = (int)((long long)nkb * split / ksplit), kb1 = (int)((long long)nkb * (split + 1) / ksplit);
  const int nblk = kb1 - kb0;
  const size_t bh0 = (size_t)b * h + hg * R4_HPG;

  // K fragments of this wave's 32 columns, all four heads: registers for the whole kernel
  Frag<__bf16> fk[R4_HPG][4];
  {
    const int jrow = min(j0 + li, N - 1);
#pragma unroll
    for (int hh = 0; hh < R4_HPG; ++hh)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fk[hh][ks].load16B(k + ((bh0 + hh) * Npad + jrow) * HD + ks * 16 + half * 8);
  }
  // loader: per block wave w moves the 4 KiB of head w's Q fragments, the 2 KiB of R block w and head w's 32 lse values
  // (7 LDS-DMA instructions).  Q and R are fragment-major: a piece is 1 KiB contiguous, lane l takes bytes 16 l, so the
  // address is a wave-uniform base (scalar registers, advanced per block) + one loop-invariant lane offset.
  const int ibw = min(wave, nib - 1);
  const unsigned smem_u = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  const unsigned voff = lane * 16;
  const char* q_w = reinterpret_cast<const char*>(q) + ((bh0 + wave) * (size_t)(Npad >> 5)) * 4096;   // [row block][k16 step][1 KiB]
  const char* r_w = reinterpret_cast<const char*>(rf_in) + rf_frag(b, nkb, 0, ibw, 0, 0) * sizeof(__bf16);
  const char* l_w = reinterpret_cast<const char*>(lse + (bh0 + wave) * N);
  auto stage = [&](int kb, int buf) {
    const unsigned base = smem_u + buf * R4_STAGE;
    const char* qs = q_w + (size_t)kb * 4096;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) r4_dma16(voff, qs + ks * 1024, base + (wave * 4 + ks) * 1024);
    if (NIB == 4 || wave == 0) {                         // NIB == 1: only R block 0 exists, wave 0 brings it
      const char* rs = r_w + (size_t)kb * (4 * 2 * 1024);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) r4_dma16(voff, rs + s2 * 1024, base + R4_QBYTES + (wave * 2 + s2) * 1024);
    }
    // lse of head `wave`, rows of the block (an ordinary load here would make hipcc drain vmcnt(0) every iteration)
    r4_dma4((unsigned)min(kb * 32 + li, N - 1) * 4u, l_w, base + R4_LSE + wave * 256);
  };

  f32x16 acc[NIB];
#pragma unroll
  for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ib][r] = 0.0f;
  const float inv_h = 1.0f / (float)h;
  const unsigned lbase = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;

  if (nblk > 0) stage(kb0, 0);
  if (R4_NSTAGE > 2 && nblk > 1) stage(kb0 + 1, 1);

  for (int it = 0; it < nblk; ++it) {
    const int kb = kb0 + it, buf = it % R4_NSTAGE;
    if (AS_ROLLOUT_ABLATE != 11) {
      // my pieces of block `it` have landed: all but the 7 (5 for the waves that carry no R block) of block it+1
      if (R4_NSTAGE == 2 || it + 1 >= nblk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (NIB == 4 || wave == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    }
    {
      // the lse rows this wave brought (head `wave`, lane l = row l % 32) become -log2(e) * lse IN PLACE: the scores'
      // accumulators start from them (below), so the term costs no MFMA and no per-score VALU
      float own;
      const unsigned a_own = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + buf * R4_STAGE + R4_LSE + wave * 256 + lane * 4;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(own) : "v"(a_own) : "memory");
      own *= -LOG2E;
      asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(a_own), "v"(own) : "memory");
    }
    if (AS_ROLLOUT_ABLATE != 14) __builtin_amdgcn_s_barrier();   // block `it` is complete; everyone is done with block it-1
    if (AS_ROLLOUT_ABLATE != 11 && it + R4_NSTAGE - 1 < nblk) stage(kb + R4_NSTAGE - 1, (it + R4_NSTAGE - 1) % R4_NSTAGE);
    const unsigned sb = lbase + buf * R4_STAGE;
    // -log2(e) * lse of the block's 32 rows in ACCUMULATOR layout: register r of a lane in half `half` is row
    // (r & 3) + 8 (r >> 2) + 4 half, i.e. four runs of 4 consecutive rows = four 16-byte (broadcast) reads per head
    const unsigned cb = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + buf * R4_STAGE + R4_LSE + half * 16;
    r4_u32x4 fc[2][4];
    auto read_c = [&](int hh, r4_u32x4 (&c)[4]) {
      const unsigned a = cb + hh * 256;
      r4_lds_read128_dyn(c[0], a);
      r4_lds_read128<32>(c[1], a);
      r4_lds_read128<64>(c[2], a);
      r4_lds_read128<96>(c[3], a);
    };
    f32x16 pbar;
#pragma unroll
    for (int r = 0; r < 16; ++r) pbar[r] = 0.0f;
    // software pipeline over the heads: the q.k MFMAs of head hh+1 are issued before the exp2 pass of head hh, the R
    // fragments are fetched under the last head's exp2 pass, and the eight R . Pbar MFMAs run as four independent chains
    r4_u32x4 fa[2][4];
    read_c(0, fc[0]);
    r4_lds_read128_dyn(fa[0][0], sb);
    r4_lds_read128<1024>(fa[0][1], sb);
    r4_lds_read128<2048>(fa[0][2], sb);
    r4_lds_read128<3072>(fa[0][3], sb);
    {
      const unsigned a1 = sb + 4096;
      read_c(1, fc[1]);
      r4_lds_read128_dyn(fa[1][0], a1);
      r4_lds_read128<1024>(fa[1][1], a1);
      r4_lds_read128<2048>(fa[1][2], a1);
      r4_lds_read128<3072>(fa[1][3], a1);
    }
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fc[0][0]),
                 "+v"(fc[0][1]), "+v"(fc[0][2]), "+v"(fc[0][3]));
    __builtin_amdgcn_sched_barrier(0);
    auto qk = [&](int hh, r4_u32x4 (&f)[4], r4_u32x4 (&c)[4]) {
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = __uint_as_float(c[r >> 2][r & 3]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<__bf16> fq;
        fq.v = *reinterpret_cast<bf16x8*>(&f[ks]);
        if (AS_ROLLOUT_ABLATE != 13) sc = mma32(fq, fk[hh][ks], sc);
        else sc[ks] += (float)fq.v[0];
      }
      return sc;
    };
    f32x16 sc_cur = qk(0, fa[0], fc[0]);
    r4_u32x4 fr[NIB][2];
#pragma unroll
    for (int hh = 0; hh < R4_HPG; ++hh) {
      f32x16 sc_next;
      if (hh + 1 < R4_HPG) {
        const int nx = (hh + 1) & 1;
        r4_wait_lds8(fa[nx][0], fa[nx][1], fa[nx][2], fa[nx][3], fc[nx][0], fc[nx][1], fc[nx][2], fc[nx][3]);
        sc_next = qk(hh + 1, fa[nx], fc[nx]);
        if (hh + 2 < R4_HPG) {                           // head hh+2's fragments into the buffer head hh just released
          const unsigned a2 = sb + (hh + 2) * 4096;
          read_c(hh + 2, fc[hh & 1]);
          r4_lds_read128_dyn(fa[hh & 1][0], a2);
          r4_lds_read128<1024>(fa[hh & 1][1], a2);
          r4_lds_read128<2048>(fa[hh & 1][2], a2);
          r4_lds_read128<3072>(fa[hh & 1][3], a2);
        }
      } else {
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) {
          const unsigned a = sb + R4_QBYTES + min(ib, nib - 1) * 2048;
          r4_lds_read128_dyn(fr[ib][0], a);
          r4_lds_read128<1024>(fr[ib][1], a);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) pbar[r] += AS_ROLLOUT_ABLATE == 12 ? sc_cur[r] : __builtin_amdgcn_exp2f(sc_cur[r]);
      if (hh + 1 < R4_HPG) sc_cur = sc_next;
    }
    Frag<__bf16> fp[2];
    if (kb * 32 + 32 <= N) {                             // (wave-uniform) every contraction row of the block exists
#pragma unroll
      for (int r = 0; r < 16; ++r) fp[r >> 3].set(r & 7, pbar[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) fp[r >> 3].set(r & 7, kb * 32 + acc_row(r, half) < N ? pbar[r] : 0.0f);
    }
    if constexpr (NIB == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0][0]), "+v"(fr[0][1]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0][0]), "+v"(fr[0][1]), "+v"(fr[1][0]), "+v"(fr[1][1]), "+v"(fr[2][0]),
                   "+v"(fr[2][1]), "+v"(fr[3][0]), "+v"(fr[3][1]));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int ib = 0; ib < NIB; ++ib)
        if (ib < nib) {
          Frag<__bf16> f;
          f.v = *reinterpret_cast<bf16x8*>(&fr[ib][s2]);
          if (AS_ROLLOUT_ABLATE != 15) acc[ib] = mma32(f, fp[s2], acc[ib]);
          else acc[ib][0] += (float)f.v[0] + (float)fp[s2].v[0];
        }
  }

  const int pidx = split * ngroups + hg;
  const int j = j0 + li;
  if (j < N) {
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = ib * 32 + acc_row(r, half);
        if (i < Trows) part[(((size_t)pidx * B + b) * Trows + i) * N + j] = acc[ib][r] * inv_h;   // the head MEAN
      }
  }
}

// AS_ROLLOUT_V3=1 (debug knob): take rollout_step3 where rollout_step4 would run; read ONCE per process
bool rollout_force_v3() {
  static const bool v = getenv("AS_ROLLOUT_V3") != nullptr;
  return v;
}

// partial products of rollout_step4: (contraction splits) x (head groups), at most R4_MAXPARTS; the split count is chosen
// so that the grid fills whole rounds of 2 workgroups per CU (e.g. 33 x 2 x 3 = 198 units -> 5 splits = 990 of 1024)
constexpr int R4_MAXPARTS = 16;
int rollout4_ksplit(int B, int N, int h, int slots) {
  static const int forced = [] { const char* e = getenv("AS_R4_KS"); return e ? atoi(e) : 0; }();   // (experiments)
  if (forced > 0 && forced * (h / R4_HPG) <= 16) return forced;
  const int units = as_ceil_div(N, 128) * B * (h / R4_HPG);
  int best = 1;
  float best_eff = 0.0f;
  for (int ks = 1; ks * (h / R4_HPG) <= R4_MAXPARTS && ks <= 8; ++ks) {
    const int wgs = units * ks;
    const float eff = (float)wgs / (float)(as_ceil_div(wgs, slots) * slots);
    if (eff > best_eff + 0.02f) { best_eff = eff; best = ks; }
  }
  return best;
}

// R_out = 0.5 (sum of the contraction-split partials, in split order + R_in), plus the fragment-major copy (all 128 x
// nkb*32 slots, zeros outside [Trows) x [N)).  grid (nkb*32/64, 128, B), 64 threads along j.
template <typename T>
__global__ __launch_bounds__(64) void rollout_finish_kernel(const float* __restrict__ part, const float* __restrict__ Rin,
                                                            float* __restrict__ Rout, T* __restrict__ rf_out, int B, int N,
                                                            int Trows, int nsplit) {
  const int nkb = (N + 31) / 32;
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j >= nkb * 32) return;
  float v = 0.0f;
  if (i < Trows && j < N) {
    const size_t idx = ((size_t)b * Trows + i) * N + j;
    float a = 0.0f;
    for (int sp = 0; sp < nsplit; ++sp) a += part[(size_t)sp * B * Trows * N + idx];
    v = 0.5f * (a + Rin[idx]);
    Rout[idx] = v;
  }
  if (rf_out != nullptr) rf_out[rf_slot(b, nkb, i, j)] = from_f32<T>(v);
}

template <typename T>
int launch_rollout_step2(const void* q, const void* k, const float* lse, const float* Rin, const void* rf_in, float* Rout,
                         void* rf_out, float* part, int nsplit, int B, int N, int h, int Trows, hipStream_t s) {
  const int Npad = as_round_up(N, 64);
  const int hpw = as_ceil_div(h, 4);
  dim3 grid(as_ceil_div(N, 32), B, nsplit);
  const size_t lds = (size_t)h * 32 * Kj2<T>::PITCH + 2 * 4 * 4 * 64 * sizeof(float4);
  AS_REQUIRE(lds <= 160 * 1024, AS_E_UNSUPPORTED, "rollout_step: LDS %zu B exceeds 160 KiB (h=%d)", lds, h);
#define AS_RO2(HPW)                                                                                            \
  do {                                                                                                         \
    (void)hipFuncSetAttribute((const void*)rollout_step2_kernel<T, HPW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                       \
    hipLaunchKernelGGL((rollout_step2_kernel<T, HPW>), grid, dim3(RO_NT), lds, s, (const T*)q, (const T*)k, lse, Rin, \
                       (const T*)rf_in, Rout, (T*)rf_out, part, B, N, Npad, h, Trows, nsplit);                 \
  } while (0)
  if (sizeof(T) == 2 && h % R4_HPG == 0 && part != nullptr && nsplit == R4_MAXPARTS && !rollout_force_v3()) {
    // bf16, heads in groups of four: streamed-operand kernel; `nsplit` here only says that the workspace holds
    // R4_MAXPARTS partial products
    const int ng = h / R4_HPG;
    // (512 slots for both forms: with three resident workgroups per CU the 32-row form still measures best at the split that
    // fills 512 -- 990 workgroups at config 2: 0.650 ms for the 6 steps against 0.72 at the 594 a 768-slot model picks)
    const int ks = rollout4_ksplit(B, N, h, 512);
    const size_t lds4 = (size_t)(Trows <= 32 ? R4Cfg<1>::NSTAGE : R4Cfg<4>::NSTAGE) * R4_STAGE;
    if (Trows <= 32) {
      (void)hipFuncSetAttribute((const void*)rollout_step4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
      hipLaunchKernelGGL(rollout_step4_kernel<1>, dim3(as_ceil_div(N, 128), B, ks * ng), dim3(RO_NT), lds4, s, (const __bf16*)q,
                         (const __bf16*)k, lse, (const __bf16*)rf_in, part, B, N, Npad, h, Trows, ks);
    } else {
      (void)hipFuncSetAttribute((const void*)rollout_step4_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
      hipLaunchKernelGGL(rollout_step4_kernel<4>, dim3(as_ceil_div(N, 128), B, ks * ng), dim3(RO_NT), lds4, s, (const __bf16*)q,
                         (const __bf16*)k, lse, (const __bf16*)rf_in, part, B, N, Npad, h, Trows, ks);
    }
    AS_CHECK_LAUNCH("rollout_step4");
    dim3 fg(as_ceil_div(as_ceil_div(N, 32) * 32, 64), as_round_up(Trows, 32), B);   // rows of the R blocks that exist
    hipLaunchKernelGGL((rollout_finish_kernel<T>), fg, dim3(64), 0, s, (const float*)part, Rin, Rout, (T*)rf_out, B, N,
                       Trows, ks * ng);
    AS_CHECK_LAUNCH("rollout_finish");
    return AS_OK;
  }
  if (sizeof(T) == 2 && h >= 8 && (getenv("AS_ROLLOUT_V2") == nullptr)) {
    // bf16, 8..16 heads (K tile 32..64 KiB): barrier-free kernel, LDS = K tile only
    const size_t lds3 = (size_t)h * 32 * Kj2<T>::PITCH;
    (void)hipFuncSetAttribute((const void*)rollout_step3_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    hipLaunchKernelGGL((rollout_step3_kernel<T>), grid, dim3(RO_NT), lds3, s, (const T*)q, (const T*)k, lse, Rin,
                       (const T*)rf_in, Rout, (T*)rf_out, part, B, N, Npad, h, Trows, nsplit);
  } else {
  switch (hpw) {
    case 1: AS_RO2(1); break;
    case 2: AS_RO2(2); break;
    case 3: AS_RO2(3); break;
    case 4: AS_RO2(4); break;
    default: AS_REQUIRE(false, AS_E_UNSUPPORTED, "rollout_step: h=%d heads (max 16)", h);
  }
  }
#undef AS_RO2
  AS_CHECK_LAUNCH("rollout_step2");
  if (nsplit > 1) {
    dim3 fg(as_ceil_div(as_ceil_div(N, 32) * 32, 64), as_round_up(Trows, 32), B);   // rows of the R blocks that exist
    hipLaunchKernelGGL((rollout_finish_kernel<T>), fg, dim3(64), 0, s, (const float*)part, Rin, Rout, (T*)rf_out, B, N,
                       Trows, nsplit);
    AS_CHECK_LAUNCH("rollout_finish");
  }
  return AS_OK;
}

// contraction split of a roll-out step: enough workgroups for >= 8 per CU-round granularity, bounded partial buffers
int rollout_nsplit(int B, int N) {
  const int wgs = as_ceil_div(N, 32) * B;
  int ns = 1;
  while (ns < 16 && wgs * ns < 2048) ns *= 2;
  return wgs >= 2048 ? 1 : ns;
}

template <typename T>
int launch_mean_rows(const void* q, const void* k, const float* lse, float* out, void* rf_out, int B, int N, int h,
                     int row0, int nrows, bool top, hipStream_t s) {
  const int Npad = as_round_up(N, 64);
  dim3 grid(as_ceil_div(as_ceil_div(N, 32), 4), as_ceil_div(nrows, 32), B);
  if (top)
    hipLaunchKernelGGL((attn_mean_rows_kernel<T, true>), grid, dim3(RO_NT), 0, s, (const T*)q, (const T*)k, lse,
                       out, (T*)rf_out, B, N, Npad, h, row0, nrows);
  else
    hipLaunchKernelGGL((attn_mean_rows_kernel<T, false>), grid, dim3(RO_NT), 0, s, (const T*)q, (const T*)k, lse,
                       out, (T*)rf_out, B, N, Npad, h, row0, nrows);
  AS_CHECK_LAUNCH("attn_mean_rows");
  return AS_OK;
}

}  // namespace

extern "C" size_t as_rollout_rfrag_bytes(int B, int N, int dtype) {
  if (B <= 0 || N <= 0) return 0;
  return (size_t)B * as_ceil_div(N, 32) * 4 * 2 * 64 * 8 * (dtype == AS_BF16 ? 2 : 4);
}

extern "C" int as_attn_mean_rows(const void* q, const void* k, const float* lse, float* out, int B, int N, int h,
                                 int row0, int nrows, int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && lse && out, AS_E_BADARG, "as_attn_mean_rows: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && h > 0 && row0 >= 0 && nrows > 0 && row0 + nrows <= N, AS_E_BADARG,
             "as_attn_mean_rows: bad row range %d+%d of %d", row0, nrows, N);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_mean_rows<__bf16>(q, k, lse, out, nullptr, B, N, h, row0, nrows, false, s);
  if (dtype == AS_F32) return launch_mean_rows<float>(q, k, lse, out, nullptr, B, N, h, row0, nrows, false, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_attn_mean_rows: dtype %d", dtype);
}

extern "C" int as_rollout_top(const void* q, const void* k, const float* lse, float* R_out, void* rf_out, int B, int N,
                              int h, int T, int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && lse && R_out, AS_E_BADARG, "as_rollout_top: null pointer");
  AS_REQUIRE(B > 0 && h > 0 && T > 0 && T <= N && T <= 128, AS_E_BADARG, "as_rollout_top: bad sizes N=%d T=%d", N, T);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_mean_rows<__bf16>(q, k, lse, R_out, rf_out, B, N, h, N - T, T, true, s);
  if (dtype == AS_F32) return launch_mean_rows<float>(q, k, lse, R_out, rf_out, B, N, h, N - T, T, true, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_rollout_top: dtype %d", dtype);
}

namespace {
// fragment-major copy of a row-major R [B, Trows, N] (zeros outside): the `rf` operand of as_rollout_step
template <typename T>
__global__ __launch_bounds__(64) void rollout_pack_kernel(const float* __restrict__ R, T* __restrict__ rf, int B, int N, int Trows) {
  const int nkb = (N + 31) / 32;
  const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j >= nkb * 32) return;
  const float v = (i < Trows && j < N) ? R[((size_t)b * Trows + i) * N + j] : 0.0f;
  rf[rf_slot(b, nkb, i, j)] = from_f32<T>(v);
}
}  // namespace

extern "C" int as_rollout_pack(const float* R, void* rf, int B, int N, int T, int dtype, as_stream_t stream) {
  AS_REQUIRE(R && rf, AS_E_BADARG, "as_rollout_pack: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && T > 0 && T <= 128, AS_E_BADARG, "as_rollout_pack: bad sizes N=%d T=%d", N, T);
  dim3 grid(as_ceil_div(as_ceil_div(N, 32) * 32, 64), as_round_up(T, 32), B);
  if (dtype == AS_BF16)
    hipLaunchKernelGGL(rollout_pack_kernel<__bf16>, grid, dim3(64), 0, (hipStream_t)stream, R, (__bf16*)rf, B, N, T);
  else if (dtype == AS_F32)
    hipLaunchKernelGGL(rollout_pack_kernel<float>, grid, dim3(64), 0, (hipStream_t)stream, R, (float*)rf, B, N, T);
  else
    AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_rollout_pack: dtype %d", dtype);
  AS_CHECK_LAUNCH("rollout_pack");
  return AS_OK;
}

extern "C" size_t as_rollout_step_workspace_bytes(int B, int N, int T) {
  if (B <= 0 || N <= 0 || T <= 0) return 0;
  // room for the R4_MAXPARTS partial products of rollout_step4 (>= the contraction splits of the older kernels)
  return (size_t)R4_MAXPARTS * B * T * N * sizeof(float);
}

extern "C" int as_rollout_step(const void* q, const void* k, const float* lse, const float* R_in, const void* rf_in,
                               float* R_out, void* rf_out, void* workspace, size_t workspace_bytes, int B, int N, int h,
                               int T, int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && lse && R_in && rf_in && R_out && R_in != R_out && rf_in != rf_out, AS_E_BADARG,
             "as_rollout_step: null/aliased pointer");
  AS_REQUIRE(B > 0 && h > 0 && h <= 16 && T > 0 && T <= N && T <= 128, AS_E_BADARG,
             "as_rollout_step: bad sizes N=%d T=%d h=%d", N, T, h);
  hipStream_t s = (hipStream_t)stream;
  // without a workspace the step runs unsplit (one workgroup per 32-column block and image)
  const size_t need = as_rollout_step_workspace_bytes(B, N, T);
  const bool have_ws = workspace != nullptr && workspace_bytes >= need && need > 0;
  const bool stream4 = have_ws && dtype == AS_BF16 && h % R4_HPG == 0 && !rollout_force_v3();
  const int ns = stream4 ? R4_MAXPARTS : (have_ws ? rollout_nsplit(B, N) : 1);
  float* part = ns > 1 ? (float*)workspace : nullptr;
  if (dtype == AS_BF16)
    return launch_rollout_step2<__bf16>(q, k, lse, R_in, rf_in, R_out, rf_out, part, ns, B, N, h, T, s);
  if (dtype == AS_F32) return launch_rollout_step2<float>(q, k, lse, R_in, rf_in, R_out, rf_out, part, ns, B, N, h, T, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_rollout_step: dtype %d", dtype);
}
