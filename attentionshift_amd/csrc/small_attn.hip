// Small-N batched multi-head self-attention for gfx950 (the MAE-decoder box / mask heads run thousands of
// 50- / 197-token attention problems of head dim 32 per step -- models/vision_transformer.py:62-86 `Attention.forward`
// as used by mmdet/models/roi_heads/bbox_heads/mae_bbox_head_rec.py:148-168 and mask_heads/mae_mask_head_pointSup.py).
//
// At these sizes a problem-head is a few hundred KFLOP: the tiles of the long-sequence kernel (sdpa.hip: 128 queries x
// 64 keys per workgroup, K / V through an LDS ring) would be mostly padding and staging, so here a single WAVE takes a
// (problem, head, 32-query block), reads its operands straight from the packed qkv tensor the reference's
// `qkv(x).reshape(B, N, 3, h, d)` produces (fp32 or bf16) and keeps everything else in registers; fp32 accumulation.
//
//   forward   out[b,n,h*d] = softmax(q k^T d^-0.5) v,  lse[b,h,n] (natural log) kept for the backward
//   backward  recomputes P from q, k, lse (no [N,N] tensor): one MFMA kernel with a lane per query -> dq (+ delta),
//             one with a lane per key -> dk, dv.  No atomics, fixed summation order.
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

// ---- backward: two MFMA kernels in the same arrangement, probabilities recomputed from q, k, lse ---------------
// gather 8 values `stride` apart (the transposed operand of the second product of each kernel)
template <typename T> __device__ __forceinline__ void gather8(Frag<T>& f, const T* p, int first, int limit, size_t stride) {
#pragma unroll
  for (int t = 0; t < 8; ++t) f.v[t] = p[(size_t)min(first + t, limit) * stride];
}

// dq: wave task = (query block, head); lane owns a query.  Also writes delta[b,h,q] = dO_q . O_q for the dk/dv pass.
//   S^T = K Q^T,  dP^T = V dO^T (same accumulator layout),  dS^T = P^T (dP^T - delta_q),  dQ^T = K^T dS
template <typename T>
__global__ __launch_bounds__(SA_NT) void small_attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                                  const T* __restrict__ d_out, const float* __restrict__ lse,
                                                                  T* __restrict__ dqkv, float* __restrict__ delta, int N,
                                                                  int h, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nqb = (N + 31) / 32;
  const int task = blockIdx.x * 4 + wave, b = blockIdx.y;
  if (task >= nqb * h) return;
  const int hh = task / nqb, qb = task - hh * nqb;
  const int li = lane & 31, half = lane >> 5;
  const size_t rs = (size_t)3 * h * SA_D, os = (size_t)h * SA_D;
  const T* base = qkv + (size_t)b * N * rs + (size_t)hh * SA_D;
  const int q_row = min(qb * 32 + li, N - 1);
  const T* orow = out + ((size_t)b * N + q_row) * os + (size_t)hh * SA_D;
  const T* grow = d_out + ((size_t)b * N + q_row) * os + (size_t)hh * SA_D;
  Frag<T> fq[2], fg[2];
  float dl = 0.0f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    fq[ks].load16B(base + (size_t)q_row * rs + ks * 16 + half * 8);
    fg[ks].load16B(grow + ks * 16 + half * 8);
    Frag<T> fo;
    fo.load16B(orow + ks * 16 + half * 8);
#pragma unroll
    for (int t = 0; t < 8; ++t) dl = fmaf(to_f32<T>(fg[ks].v[t]), to_f32<T>(fo.v[t]), dl);
  }
  dl += __shfl_xor(dl, 32);                                   // the halves hold complementary head-dim slices
  const float l2 = lse[((size_t)b * h + hh) * N + q_row] * LOG2E;
  const int pi = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);
  const float c2 = scale * LOG2E;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int k0 = 0; k0 < N; k0 += 32) {
    const int k_row = min(k0 + pi, N - 1);
    Frag<T> fk[2], fv[2], fkt[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      fk[ks].load16B(base + (size_t)k_row * rs + (size_t)h * SA_D + ks * 16 + half * 8);
      fv[ks].load16B(base + (size_t)k_row * rs + (size_t)2 * h * SA_D + ks * 16 + half * 8);
      gather8<T>(fkt[ks], base + (size_t)h * SA_D + li, k0 + 16 * ks + 8 * half, N - 1, rs);       // K^T: row = d index
    }
    f32x16 sc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = 0.0f; dp[r] = 0.0f; }
    sc = mma32(fk[0], fq[0], sc); sc = mma32(fk[1], fq[1], sc);
    dp = mma32(fv[0], fg[0], dp); dp = mma32(fv[1], fg[1], dp);
    Frag<T> fs[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + 8 * half + (r & 7) + 16 * (r >> 3);
      const float p = key < N ? __builtin_amdgcn_exp2f(sc[r] * c2 - l2) : 0.0f;
      fs[r >> 3].set(r & 7, p * (dp[r] - dl));
    }
    acc = mma32(fkt[0], fs[0], acc);
    acc = mma32(fkt[1], fs[1], acc);
  }
  const int q = qb * 32 + li;
  if (q < N) {
    T* o = dqkv + ((size_t)b * N + q) * rs + (size_t)hh * SA_D;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[acc_row(r, half)] = from_f32<T>(acc[r] * scale);
    if (half == 0) delta[((size_t)b * h + hh) * N + q] = dl;
  }
}

// dk, dv: wave task = (key block, head); lane owns a key, registers run over 8 consecutive queries.
//   S = Q K^T,  dP = dO V^T,  dV^T = dO^T P,  dK^T = Q^T dS * scale
template <typename T>
__global__ __launch_bounds__(SA_NT) void small_attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ d_out,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ delta, T* __restrict__ dqkv,
                                                                   int N, int h, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nkb = (N + 31) / 32;
  const int task = blockIdx.x * 4 + wave, b = blockIdx.y;
  if (task >= nkb * h) return;
  const int hh = task / nkb, kb = task - hh * nkb;
  const int li = lane & 31, half = lane >> 5;
  const size_t rs = (size_t)3 * h * SA_D, os = (size_t)h * SA_D;
  const T* base = qkv + (size_t)b * N * rs + (size_t)hh * SA_D;
  const T* gbase = d_out + (size_t)b * N * os + (size_t)hh * SA_D;
  const float* lrow = lse + ((size_t)b * h + hh) * N;
  const float* drow = delta + ((size_t)b * h + hh) * N;
  const int k_row = min(kb * 32 + li, N - 1);
  Frag<T> fk[2], fv[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    fk[ks].load16B(base + (size_t)k_row * rs + (size_t)h * SA_D + ks * 16 + half * 8);
    fv[ks].load16B(base + (size_t)k_row * rs + (size_t)2 * h * SA_D + ks * 16 + half * 8);
  }
  const int pi = (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1);
  const float c2 = scale * LOG2E;
  f32x16 adk, adv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { adk[r] = 0.0f; adv[r] = 0.0f; }
  for (int q0 = 0; q0 < N; q0 += 32) {
    const int q_row = min(q0 + pi, N - 1);
    Frag<T> fq[2], fg[2], fqt[2], fgt[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      fq[ks].load16B(base + (size_t)q_row * rs + ks * 16 + half * 8);
      fg[ks].load16B(gbase + (size_t)q_row * os + ks * 16 + half * 8);
      gather8<T>(fqt[ks], base + li, q0 + 16 * ks + 8 * half, N - 1, rs);                        // Q^T: row = d index
      gather8<T>(fgt[ks], gbase + li, q0 + 16 * ks + 8 * half, N - 1, os);                       // dO^T
    }
    float l2[16], dl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = min(q0 + 8 * half + (r & 7) + 16 * (r >> 3), N - 1);
      l2[r] = lrow[q] * LOG2E;
      dl[r] = drow[q];
    }
    f32x16 sc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = 0.0f; dp[r] = 0.0f; }
    sc = mma32(fq[0], fk[0], sc); sc = mma32(fq[1], fk[1], sc);           // [query (pi order), key]
    dp = mma32(fg[0], fv[0], dp); dp = mma32(fg[1], fv[1], dp);
    Frag<T> fp[2], fs[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + 8 * half + (r & 7) + 16 * (r >> 3);
      const float p = q < N ? __builtin_amdgcn_exp2f(sc[r] * c2 - l2[r]) : 0.0f;
      fp[r >> 3].set(r & 7, p);
      fs[r >> 3].set(r & 7, p * (dp[r] - dl[r]));
    }
    adv = mma32(fgt[0], fp[0], adv); adv = mma32(fgt[1], fp[1], adv);
    adk = mma32(fqt[0], fs[0], adk); adk = mma32(fqt[1], fs[1], adk);
  }
  const int k = kb * 32 + li;
  if (k < N) {
    T* o = dqkv + ((size_t)b * N + k) * rs + (size_t)hh * SA_D;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[(size_t)h * SA_D + acc_row(r, half)] = from_f32<T>(adk[r] * scale);
      o[(size_t)2 * h * SA_D + acc_row(r, half)] = from_f32<T>(adv[r]);
    }
  }
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

extern "C" size_t as_small_attn_bwd_workspace_bytes(int Bp, int N, int h) {
  if (Bp <= 0 || N <= 0 || h <= 0) return 0;
  return (size_t)Bp * h * N * sizeof(float);                 // delta = dO . O per (problem, head, query)
}

extern "C" int as_small_attn_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* dqkv,
                                 void* workspace, size_t workspace_bytes, int Bp, int N, int h, int d, int dtype,
                                 as_stream_t stream) {
  AS_REQUIRE(qkv && out && d_out && lse && dqkv && workspace, AS_E_BADARG, "as_small_attn_bwd: null pointer");
  AS_REQUIRE(Bp > 0 && N > 0 && h > 0, AS_E_BADARG, "as_small_attn_bwd: bad sizes");
  AS_REQUIRE(d == SA_D && (dtype == AS_F32 || dtype == AS_BF16), AS_E_UNSUPPORTED,
             "as_small_attn_bwd: head dim %d (only %d), dtype %d", d, SA_D, dtype);
  AS_REQUIRE(workspace_bytes >= as_small_attn_bwd_workspace_bytes(Bp, N, h), AS_E_WORKSPACE,
             "as_small_attn_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)d);
  const dim3 grid(as_ceil_div(as_ceil_div(N, 32) * h, 4), Bp);
  float* delta = (float*)workspace;
  if (dtype == AS_F32) {
    hipLaunchKernelGGL((small_attn_bwd_dq_kernel<float>), grid, dim3(SA_NT), 0, s, (const float*)qkv, (const float*)out,
                       (const float*)d_out, lse, (float*)dqkv, delta, N, h, scale);
    hipLaunchKernelGGL((small_attn_bwd_dkv_kernel<float>), grid, dim3(SA_NT), 0, s, (const float*)qkv, (const float*)d_out,
                       lse, delta, (float*)dqkv, N, h, scale);
  } else {
    hipLaunchKernelGGL((small_attn_bwd_dq_kernel<__bf16>), grid, dim3(SA_NT), 0, s, (const __bf16*)qkv, (const __bf16*)out,
                       (const __bf16*)d_out, lse, (__bf16*)dqkv, delta, N, h, scale);
    hipLaunchKernelGGL((small_attn_bwd_dkv_kernel<__bf16>), grid, dim3(SA_NT), 0, s, (const __bf16*)qkv, (const __bf16*)d_out,
                       lse, delta, (__bf16*)dqkv, N, h, scale);
  }
  AS_CHECK_LAUNCH("small_attn_bwd");
  return AS_OK;
}
