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
#include "common.h"

namespace {

constexpr int HD = 64, RO_NT = 256;
constexpr float LOG2E = 1.44269504088896340736f;

template <typename T> struct KjCfg { static constexpr int PITCH = HD * (int)sizeof(T) + 16; };

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
  const float c2 = 0.125f * LOG2E;
  for (int hh = 0; hh < h; ++hh) {
    const size_t bh = (size_t)b * h + hh;
    const T* qrow = q + (bh * Npad + irow) * HD + half * 8;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag<T> fa, fb;
      fa.load16B(qrow + ks * 16);
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
      pbar[r] += __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lrow[row] * LOG2E));
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
                                                               float* __restrict__ out, int B, int N, int Npad,
                                                               int h, int row0, int nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int jb = blockIdx.x * 4 + wave, ib = blockIdx.y, b = blockIdx.z;
  const int j0 = jb * 32, i0 = ib * 32;
  if (j0 >= N) return;
  const f32x16 p = pbar_tile<T>(q, k, lse, nullptr, b, h, N, Npad, row0 + i0, j0, li, half);
  const int j = j0 + li;
  if (j < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + acc_row(r, half);
      if (i < nrows) {
        float v = p[r];
        if (TOP) v = 0.5f * (v + ((row0 + i) == j ? 1.0f : 0.0f));
        out[((size_t)b * nrows + i) * N + j] = v;
      }
    }
  }
}

__device__ __forceinline__ void load_r_frag(Frag<__bf16>& f, const float* p, bool v0, bool v1) {
#pragma unroll
  for (int t = 0; t < 4; ++t) f.v[t] = (__bf16)(v0 ? p[t] : 0.0f);
#pragma unroll
  for (int t = 0; t < 4; ++t) f.v[4 + t] = (__bf16)(v1 ? p[8 + t] : 0.0f);
}
__device__ __forceinline__ void load_r_frag(Frag<float>& f, const float* p, bool v0, bool v1) {
#pragma unroll
  for (int t = 0; t < 4; ++t) f.v[t] = v0 ? p[t] : 0.0f;
#pragma unroll
  for (int t = 0; t < 4; ++t) f.v[4 + t] = v1 ? p[8 + t] : 0.0f;
}

// R_out[b] = 0.5 * (R_in[b] . mean_h P + R_in[b]);  one workgroup = 32 output columns of one image,
// its 4 waves split the contraction (k) range and are reduced through LDS at the end.
template <typename T, int IB>   // IB = number of 32-row blocks of R (T <= 32*IB)
__global__ __launch_bounds__(RO_NT) void rollout_step_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ Rin,
                                                             float* __restrict__ Rout, int B, int N, int Npad,
                                                             int h, int Trows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int j0 = blockIdx.x * 32, b = blockIdx.y;

  // stage K rows j0..j0+31 of every head: [h][32][PITCH]
  {
    constexpr int CPR = HD * (int)sizeof(T) / 16;     // 16-byte chunks per row
    const int total = h * 32 * CPR;
    for (int c = tid; c < total; c += RO_NT) {
      const int hh = c / (32 * CPR), rem = c % (32 * CPR);
      const int jr = rem / CPR, ch = rem % CPR;
      const int jrow = min(j0 + jr, N - 1);
      const uint4 u = *reinterpret_cast<const uint4*>(
          reinterpret_cast<const char*>(k + (((size_t)b * h + hh) * Npad + jrow) * HD) + ch * 16);
      *reinterpret_cast<uint4*>(smem + ((size_t)hh * 32 + jr) * KjCfg<T>::PITCH + ch * 16) = u;
    }
  }
  __syncthreads();

  f32x16 acc[IB];
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ib][r] = 0.0f;

  const int nkb = (N + 31) / 32;
  const float* Rb = Rin + (size_t)b * Trows * N;
  for (int kb = wave; kb < nkb; kb += 4) {
    const int k0 = kb * 32;
    f32x16 p = pbar_tile<T>(q, k, lse, smem, b, h, N, Npad, k0, j0, li, half);
    // rows (contraction index) beyond N contribute nothing
    if (k0 + 32 > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (k0 + acc_row(r, half) >= N) p[r] = 0.0f;
    }
    Frag<T> fp[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) fp[r >> 3].set(r & 7, p[r]);
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
      const int i = ib * 32 + li;
      const bool iv = i < Trows;
      const float* rrow = Rb + (size_t)min(i, Trows - 1) * N;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kk = k0 + 16 * s + 4 * half;          // runs kk..kk+3 and kk+8..kk+11
        Frag<T> fr;
        if (k0 + 32 <= N) {
          load_r_frag(fr, rrow + kk, iv, iv);
        } else {                                        // ragged tail: element-wise guard
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int kx = kk + 8 * (t >> 2) + (t & 3);
            fr.set(t, (iv && kx < N) ? rrow[kx] : 0.0f);
          }
        }
        acc[ib] = mma32(fr, fp[s], acc[ib]);
      }
    }
  }

  // cross-wave reduction through LDS (aliases the K staging area)
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);          // [4 waves][IB*32 rows][32 cols]
#pragma unroll
  for (int ib = 0; ib < IB; ++ib)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      red[((size_t)wave * IB * 32 + ib * 32 + acc_row(r, half)) * 32 + li] = acc[ib][r];
  __syncthreads();
  for (int e = tid; e < IB * 32 * 32; e += RO_NT) {
    const int i = e / 32, jj = e % 32;
    const int j = j0 + jj;
    if (i < Trows && j < N) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[((size_t)w * IB * 32 + i) * 32 + jj];
      const size_t idx = ((size_t)b * Trows + i) * N + j;
      Rout[idx] = 0.5f * (v + Rin[idx]);
    }
  }
}

template <typename T>
int launch_mean_rows(const void* q, const void* k, const float* lse, float* out, int B, int N, int h, int row0,
                     int nrows, bool top, hipStream_t s) {
  const int Npad = as_round_up(N, 64);
  dim3 grid(as_ceil_div(as_ceil_div(N, 32), 4), as_ceil_div(nrows, 32), B);
  if (top)
    hipLaunchKernelGGL((attn_mean_rows_kernel<T, true>), grid, dim3(RO_NT), 0, s, (const T*)q, (const T*)k, lse,
                       out, B, N, Npad, h, row0, nrows);
  else
    hipLaunchKernelGGL((attn_mean_rows_kernel<T, false>), grid, dim3(RO_NT), 0, s, (const T*)q, (const T*)k, lse,
                       out, B, N, Npad, h, row0, nrows);
  AS_CHECK_LAUNCH("attn_mean_rows");
  return AS_OK;
}

template <typename T>
int launch_rollout_step(const void* q, const void* k, const float* lse, const float* Rin, float* Rout, int B,
                        int N, int h, int Trows, hipStream_t s) {
  const int Npad = as_round_up(N, 64);
  const int IB = as_ceil_div(Trows, 32);
  dim3 grid(as_ceil_div(N, 32), B);
  size_t lds_k = (size_t)h * 32 * KjCfg<T>::PITCH;
  size_t lds_r = (size_t)4 * IB * 32 * 32 * sizeof(float);
  size_t lds = lds_k > lds_r ? lds_k : lds_r;
  AS_REQUIRE(lds <= 160 * 1024, AS_E_UNSUPPORTED, "rollout_step: LDS %zu B exceeds 160 KiB (h=%d T=%d)", lds, h, Trows);
#define AS_RO_LAUNCH(IBV)                                                                                      \
  do {                                                                                                         \
    (void)hipFuncSetAttribute((const void*)rollout_step_kernel<T, IBV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                        (int)lds);                                                                             \
    hipLaunchKernelGGL((rollout_step_kernel<T, IBV>), grid, dim3(RO_NT), lds, s, (const T*)q, (const T*)k, lse, \
                       Rin, Rout, B, N, Npad, h, Trows);                                                       \
  } while (0)
  switch (IB) {
    case 1: AS_RO_LAUNCH(1); break;
    case 2: AS_RO_LAUNCH(2); break;
    case 3: AS_RO_LAUNCH(3); break;
    case 4: AS_RO_LAUNCH(4); break;
    default: AS_REQUIRE(false, AS_E_UNSUPPORTED, "rollout_step: T=%d > 128 point tokens unsupported", Trows);
  }
#undef AS_RO_LAUNCH
  AS_CHECK_LAUNCH("rollout_step");
  return AS_OK;
}

}  // namespace

extern "C" int as_attn_mean_rows(const void* q, const void* k, const float* lse, float* out, int B, int N, int h,
                                 int row0, int nrows, int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && lse && out, AS_E_BADARG, "as_attn_mean_rows: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && h > 0 && row0 >= 0 && nrows > 0 && row0 + nrows <= N, AS_E_BADARG,
             "as_attn_mean_rows: bad row range %d+%d of %d", row0, nrows, N);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_mean_rows<__bf16>(q, k, lse, out, B, N, h, row0, nrows, false, s);
  if (dtype == AS_F32) return launch_mean_rows<float>(q, k, lse, out, B, N, h, row0, nrows, false, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_attn_mean_rows: dtype %d", dtype);
}

extern "C" int as_rollout_top(const void* q, const void* k, const float* lse, float* R_out, int B, int N, int h,
                              int T, int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && lse && R_out, AS_E_BADARG, "as_rollout_top: null pointer");
  AS_REQUIRE(B > 0 && h > 0 && T > 0 && T <= N, AS_E_BADARG, "as_rollout_top: bad sizes N=%d T=%d", N, T);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_mean_rows<__bf16>(q, k, lse, R_out, B, N, h, N - T, T, true, s);
  if (dtype == AS_F32) return launch_mean_rows<float>(q, k, lse, R_out, B, N, h, N - T, T, true, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_rollout_top: dtype %d", dtype);
}

extern "C" int as_rollout_step(const void* q, const void* k, const float* lse, const float* R_in, float* R_out,
                               int B, int N, int h, int T, int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && lse && R_in && R_out && R_in != R_out, AS_E_BADARG, "as_rollout_step: null/aliased pointer");
  AS_REQUIRE(B > 0 && h > 0 && T > 0 && T <= N, AS_E_BADARG, "as_rollout_step: bad sizes N=%d T=%d", N, T);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_rollout_step<__bf16>(q, k, lse, R_in, R_out, B, N, h, T, s);
  if (dtype == AS_F32) return launch_rollout_step<float>(q, k, lse, R_in, R_out, B, N, h, T, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_rollout_step: dtype %d", dtype);
}
