// Small-N batched multi-head self-attention for gfx950 (the MAE-decoder box / mask heads run thousands of
// 50- / 197-token attention problems of head dim 32 per step -- models/vision_transformer.py:62-86 `Attention.forward`
// as used by mmdet/models/roi_heads/bbox_heads/mae_bbox_head_rec.py:148-168 and mask_heads/mae_mask_head_pointSup.py).
//
// One workgroup per (problem, head).  At these sizes the work is a few hundred KFLOP per workgroup and the MFMA tiles
// of the long-sequence kernel (sdpa.hip: 128 queries x 64 keys, head dim 64) would be mostly padding, so this is the
// "latency / HBM" shape: K and V of the problem-head live in LDS as fp32, a thread owns a query row (q, the output
// accumulator and the running softmax state in registers), keys are broadcast reads.  fp32 accumulation throughout;
// operands fp32 or bf16 in the packed layout the reference's `qkv(x).reshape(B, N, 3, h, d)` produces.
//
//   forward   out[b,n,h*d] = softmax(q k^T d^-0.5) v,  lse[b,h,n] (natural log) kept for the backward
//   backward  recomputes P from q, k, lse (no [N,N] tensor): phase 1, a thread per query row -> dq;
//             phase 2, a thread per key row -> dk, dv.  No atomics, fixed summation order.
#include "common.h"

namespace {

constexpr int SA_NT = 256;
constexpr int SA_D = 32;
constexpr float LOG2E = 1.4426950408889634f;

template <typename T> __device__ __forceinline__ void load_row(const T* p, float* r) {
#pragma unroll
  for (int c = 0; c < SA_D; ++c) r[c] = to_f32<T>(p[c]);
}

// ---- forward: MFMA, one wave per (problem, head, block of 32 queries), no LDS ---------------------------------
// Both products are computed transposed so that a LANE owns a query (sdpa.hip's arrangement at head dim 32):
//   S^T[key, query] = K_blk . Q^T      A = K rows (fed in the order pi(i): bits 2 and 3 of the row index swapped),
//                                      B = Q rows; lane (query, half) then holds, in accumulator registers 8s..8s+7,
//                                      the scores of the 8 CONSECUTIVE keys 16 s + 8 half + 0..7 of the block
//   O^T[d, query]   = V_blk^T . P^T    B = those registers as they are (no shuffle), A = V^T gathered per key
// so the softmax of a query is register arithmetic plus one cross-half shuffle, and P never leaves the lane.
// grid (ceil(N / 32) * h / 4 rounded up, Bp) with 4 waves per workgroup; wave task = (query block, head)
template <typename T>
__global__ __launch_bounds__(SA_NT) void small_attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                               float* __restrict__ lse, int N, int h, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nqb = (N + 31) / 32;
  const int task = blockIdx.x * 4 + wave, b = blockIdx.y;
  if (task >= nqb * h) return;
  const int hh = task / nqb, qb = task - hh * nqb;
  const int li = lane & 31, half = lane >> 5;
  const size_t rs = (size_t)3 * h * SA_D;                    // elements between consecutive tokens
  const T* base = qkv + (size_t)b * N * rs + (size_t)hh * SA_D;
  const int q_row = min(qb * 32 + li, N - 1);
  Frag<T> fq[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fq[ks].load16B(base + (size_t)q_row * rs + ks * 16 + half * 8);
  const int pi = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);          // swap bits 2 and 3
  const float c2 = scale * LOG2E;

  f32x16 acc_o;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_o[r] = 0.0f;
  float m = -INFINITY, l = 0.0f;                              // running max (log2 domain), this lane's partial sum
  for (int k0 = 0; k0 < N; k0 += 32) {
    Frag<T> fk[2], fv[2];
    const int k_row = min(k0 + pi, N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fk[ks].load16B(base + (size_t)k_row * rs + (size_t)h * SA_D + ks * 16 + half * 8);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // V^T operand: row = head-dim index li, the 8 consecutive keys k0 + 16 ks + 8 half + 0..7 (clamped; their P is 0)
      const int kv = k0 + 16 * ks + 8 * half;
      Frag<T> f;
#pragma unroll
      for (int t = 0; t < 8; ++t) f.v[t] = base[(size_t)min(kv + t, N - 1) * rs + (size_t)2 * h * SA_D + li];
      fv[ks] = f;
    }
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
    sc = mma32(fk[0], fq[0], sc);
    sc = mma32(fk[1], fq[1], sc);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + 8 * half + (r & 7) + 16 * (r >> 3);
      sc[r] = key < N ? sc[r] * c2 : -INFINITY;
      mx = fmaxf(mx, sc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));                       // the other half holds the block's other 16 keys
    const float m_new = fmaxf(m, mx);
    const float corr = __builtin_amdgcn_exp2f(m - m_new);     // 0 on the first block (m = -inf)
    l *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] *= corr;
    m = m_new;
    Frag<T> fp[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(sc[r] - m);
      l += p;
      fp[r >> 3].set(r & 7, p);
    }
    acc_o = mma32(fv[0], fp[0], acc_o);
    acc_o = mma32(fv[1], fp[1], acc_o);
  }
  l += __shfl_xor(l, 32);
  const int q = qb * 32 + li;
  if (q < N) {
    const float inv = 1.0f / l;
    T* o = out + ((size_t)b * N + q) * h * SA_D + (size_t)hh * SA_D;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[acc_row(r, half)] = from_f32<T>(acc_o[r] * inv);
    if (half == 0) lse[((size_t)b * h + hh) * N + q] = (m + log2f(l)) * 0.6931471805599453f;
  }
}

// grid (h, Bp); dynamic LDS = (4 * N * D + 2 * N) floats: Q (pre-scaled), K, V, dO, lse, delta
template <typename T>
__global__ __launch_bounds__(SA_NT) void small_attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                               const T* __restrict__ d_out, const float* __restrict__ lse,
                                                               T* __restrict__ dqkv, int N, int h, float scale) {
  extern __shared__ float sm[];
  float* Qs = sm;
  float* Ks = Qs + (size_t)N * SA_D;
  float* Vs = Ks + (size_t)N * SA_D;
  float* Gs = Vs + (size_t)N * SA_D;
  float* Ls = Gs + (size_t)N * SA_D;
  float* Ds = Ls + N;
  const int hh = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t row_stride = (size_t)3 * h * SA_D;
  const T* base = qkv + (size_t)b * N * row_stride + (size_t)hh * SA_D;
  T* dbase = dqkv + (size_t)b * N * row_stride + (size_t)hh * SA_D;
  const size_t orow = (size_t)h * SA_D;
  const T* obase = out + (size_t)b * N * orow + (size_t)hh * SA_D;
  const T* gbase = d_out + (size_t)b * N * orow + (size_t)hh * SA_D;
  for (int i = tid; i < N * SA_D; i += SA_NT) {
    const int n = i / SA_D, c = i - n * SA_D;
    Qs[i] = to_f32<T>(base[n * row_stride + c]) * scale;
    Ks[i] = to_f32<T>(base[n * row_stride + (size_t)h * SA_D + c]);
    Vs[i] = to_f32<T>(base[n * row_stride + (size_t)2 * h * SA_D + c]);
    Gs[i] = to_f32<T>(gbase[n * orow + c]);
  }
  for (int n = tid; n < N; n += SA_NT) {
    Ls[n] = lse[((size_t)b * h + hh) * N + n];
    float d = 0.0f;
#pragma unroll
    for (int c = 0; c < SA_D; ++c) d = fmaf(to_f32<T>(gbase[n * orow + c]), to_f32<T>(obase[n * orow + c]), d);
    Ds[n] = d;                                        // delta_i = dO_i . O_i
  }
  __syncthreads();
  // phase 1: thread = query row i -> dq_i = scale * sum_j P_ij (dO_i . v_j - delta_i) k_j
  for (int i = tid; i < N; i += SA_NT) {
    float q[SA_D], g[SA_D], dq[SA_D];
#pragma unroll
    for (int c = 0; c < SA_D; ++c) { q[c] = Qs[i * SA_D + c]; g[c] = Gs[i * SA_D + c]; dq[c] = 0.0f; }
    const float li = Ls[i], di = Ds[i];
    for (int j = 0; j < N; ++j) {
      const float* kj = Ks + j * SA_D;
      const float* vj = Vs + j * SA_D;
      float s = 0.0f, dp = 0.0f;
#pragma unroll
      for (int c = 0; c < SA_D; ++c) { s = fmaf(q[c], kj[c], s); dp = fmaf(g[c], vj[c], dp); }
      const float ds = __expf(s - li) * (dp - di);
#pragma unroll
      for (int c = 0; c < SA_D; ++c) dq[c] = fmaf(ds, kj[c], dq[c]);
    }
#pragma unroll
    for (int c = 0; c < SA_D; ++c) dbase[i * row_stride + c] = from_f32<T>(dq[c] * scale);
  }
  // phase 2: thread = key row j -> dv_j = sum_i P_ij dO_i ;  dk_j = sum_i P_ij (dO_i . v_j - delta_i) q_i  (q pre-scaled)
  for (int j = tid; j < N; j += SA_NT) {
    float k[SA_D], v[SA_D], dk[SA_D], dv[SA_D];
#pragma unroll
    for (int c = 0; c < SA_D; ++c) { k[c] = Ks[j * SA_D + c]; v[c] = Vs[j * SA_D + c]; dk[c] = 0.0f; dv[c] = 0.0f; }
    for (int i = 0; i < N; ++i) {
      const float* qi = Qs + i * SA_D;
      const float* gi = Gs + i * SA_D;
      float s = 0.0f, dp = 0.0f;
#pragma unroll
      for (int c = 0; c < SA_D; ++c) { s = fmaf(qi[c], k[c], s); dp = fmaf(gi[c], v[c], dp); }
      const float p = __expf(s - Ls[i]);
      const float ds = p * (dp - Ds[i]);
#pragma unroll
      for (int c = 0; c < SA_D; ++c) { dv[c] = fmaf(p, gi[c], dv[c]); dk[c] = fmaf(ds, qi[c], dk[c]); }
    }
#pragma unroll
    for (int c = 0; c < SA_D; ++c) {
      dbase[j * row_stride + (size_t)h * SA_D + c] = from_f32<T>(dk[c]);
      dbase[j * row_stride + (size_t)2 * h * SA_D + c] = from_f32<T>(dv[c]);
    }
  }
}

template <typename K> int set_lds(K kern, size_t bytes) {
  if (bytes > 48 * 1024) return (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return 0;
}

}  // namespace

extern "C" int as_small_attn_fwd(const void* qkv, void* out, float* lse, int Bp, int N, int h, int d, int dtype,
                                 as_stream_t stream) {
  AS_REQUIRE(qkv && out && lse, AS_E_BADARG, "as_small_attn_fwd: null pointer");
  AS_REQUIRE(Bp > 0 && N > 0 && h > 0, AS_E_BADARG, "as_small_attn_fwd: bad sizes");
  AS_REQUIRE(d == SA_D && (dtype == AS_F32 || dtype == AS_BF16), AS_E_UNSUPPORTED,
             "as_small_attn_fwd: head dim %d (only %d), dtype %d", d, SA_D, dtype);
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)d);
  const dim3 grid(as_ceil_div(as_ceil_div(N, 32) * h, 4), Bp);
  if (dtype == AS_F32) {
    hipLaunchKernelGGL((small_attn_fwd_kernel<float>), grid, dim3(SA_NT), 0, s, (const float*)qkv, (float*)out, lse, N, h,
                       scale);
  } else {
    hipLaunchKernelGGL((small_attn_fwd_kernel<__bf16>), grid, dim3(SA_NT), 0, s, (const __bf16*)qkv, (__bf16*)out, lse, N, h,
                       scale);
  }
  AS_CHECK_LAUNCH("small_attn_fwd");
  return AS_OK;
}

extern "C" int as_small_attn_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* dqkv, int Bp,
                                 int N, int h, int d, int dtype, as_stream_t stream) {
  AS_REQUIRE(qkv && out && d_out && lse && dqkv, AS_E_BADARG, "as_small_attn_bwd: null pointer");
  AS_REQUIRE(Bp > 0 && N > 0 && h > 0, AS_E_BADARG, "as_small_attn_bwd: bad sizes");
  AS_REQUIRE(d == SA_D && (dtype == AS_F32 || dtype == AS_BF16), AS_E_UNSUPPORTED,
             "as_small_attn_bwd: head dim %d (only %d), dtype %d", d, SA_D, dtype);
  const size_t lds = ((size_t)4 * N * SA_D + 2 * N) * sizeof(float);
  AS_REQUIRE(lds <= 150 * 1024, AS_E_UNSUPPORTED, "as_small_attn_bwd: N=%d tokens exceed the LDS-resident form (<= 295)", N);
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)d);
  if (dtype == AS_F32) {
    set_lds(small_attn_bwd_kernel<float>, lds);
    hipLaunchKernelGGL((small_attn_bwd_kernel<float>), dim3(h, Bp), dim3(SA_NT), lds, s, (const float*)qkv, (const float*)out,
                       (const float*)d_out, lse, (float*)dqkv, N, h, scale);
  } else {
    set_lds(small_attn_bwd_kernel<__bf16>, lds);
    hipLaunchKernelGGL((small_attn_bwd_kernel<__bf16>), dim3(h, Bp), dim3(SA_NT), lds, s, (const __bf16*)qkv,
                       (const __bf16*)out, (const __bf16*)d_out, lse, (__bf16*)dqkv, N, h, scale);
  }
  AS_CHECK_LAUNCH("small_attn_bwd");
  return AS_OK;
}
