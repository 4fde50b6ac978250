// Fused scaled-dot-product attention forward for gfx950 (head dim 64), never materialising [h,N,N].
//
// Replaces  attn = softmax(q k^T * d^-0.5); x = attn @ v   (reference models/vision_transformer.py:79-83)
// and additionally emits the row log-sum-exp so that any attention row can be recomputed later
// (roll-out, rollout.hip) instead of returning the [B,h,N,N] matrix the reference returns.
//
// Workgroup = 4 waves = 128 query rows of one (image, head); KV tile = 64 keys.
// Both MFMAs are "swapped" so that a lane owns ONE query (column) across the whole pipeline:
//   S^T[key][query] = K . Q^T      A = K tile (LDS, [key][d] rows padded +16 B), B = Q^T (registers)
//   O^T[d][query]   = V^T . P^T    A = V^T tile (LDS, [d][key]; V is stored transposed by the QKV
//                                  epilogue), B = P^T taken straight from the S^T accumulators
// The S^T accumulator of lane (query, half) holds keys (r&3)+8(r>>2)+4*half; the P.V MFMA only
// needs A and B to agree on the key order, so the V^T fragment is read in that same permuted order
// (two runs of 4 keys per k16 step) and P never moves between lanes.  Row max / sum are in-lane
// reductions plus one cross-half exchange.  Online softmax in the exp2 domain (v_exp_f32).
#include "common.h"

namespace {

constexpr int SD_NT = 256, SD_QB = 128, SD_KB = 64, HD = 64;

template <typename T> struct SdpaCfg;
template <> struct SdpaCfg<__bf16> {
  static constexpr int K_PITCH = HD * 2 + 16;    // bytes
  static constexpr int V_PITCH = SD_KB * 2 + 8;  // 136: conflict-free ds_read_b64 down the d rows
};
template <> struct SdpaCfg<float> {
  static constexpr int K_PITCH = HD * 4 + 16;
  static constexpr int V_PITCH = SD_KB * 4 + 16;
};

// V^T fragment for (key block kb, sub-step s): element t <-> key kb*32 + 16 s + 8 (t>>2) + 4 half + (t&3)
__device__ __forceinline__ void load_vt_frag(Frag<__bf16>& f, const char* row, int key0) {
  const uint2 a = *reinterpret_cast<const uint2*>(row + key0 * 2);
  const uint2 b = *reinterpret_cast<const uint2*>(row + (key0 + 8) * 2);
  uint4 u = make_uint4(a.x, a.y, b.x, b.y);
  f.v = *reinterpret_cast<bf16x8*>(&u);
}
__device__ __forceinline__ void load_vt_frag(Frag<float>& f, const char* row, int key0) {
  const float4 a = *reinterpret_cast<const float4*>(row + key0 * 4);
  const float4 b = *reinterpret_cast<const float4*>(row + (key0 + 8) * 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}

__device__ __forceinline__ void store4(__bf16* p, float a, float b, float c, float d) {
  bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  *reinterpret_cast<bf16x4*>(p) = v;
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

template <typename T>
__global__ __launch_bounds__(SD_NT) void sdpa_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ vt, T* __restrict__ o,
                                                         float* __restrict__ lse, int B, int N, int Npad, int h) {
  using Cfg = SdpaCfg<T>;
  constexpr int EPC = 16 / (int)sizeof(T);                   // elements per 16-byte chunk
  constexpr int K_CHUNKS = SD_KB * HD / EPC / SD_NT;         // per thread
  constexpr int V_CHUNKS = HD * SD_KB / EPC / SD_NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + SD_KB * Cfg::K_PITCH;

  const int BH = B * h;
  const int bid = blockIdx.x;
  const int bh = bid % BH, qt = bid / BH;         // consecutive blocks (= XCDs) take different heads
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int query = qt * SD_QB + wave * 32 + li;
  const int qclamped = min(query, N - 1);

  const T* kb_ = k + (size_t)bh * Npad * HD;
  const T* vb = vt + (size_t)bh * HD * Npad;

  Frag<T> fq[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fq[ks].load16B(q + qf_frag((size_t)bh, Npad, qclamped, ks, half));

  uint4 rk[K_CHUNKS], rv[V_CHUNKS];
  const int nkt = Npad / SD_KB;
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < K_CHUNKS; ++i) {
      const int c = tid + i * SD_NT;      // K tile is one contiguous block of 64 rows x 64 elements
      rk[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(kb_ + (size_t)kt * SD_KB * HD) + (size_t)c * 16);
    }
#pragma unroll
    for (int i = 0; i < V_CHUNKS; ++i) {
      const int c = tid + i * SD_NT;
      const int d = c / (SD_KB / EPC), ch = c % (SD_KB / EPC);
      const int key = kt * SD_KB + ch * EPC;
      uint4 u = *reinterpret_cast<const uint4*>(vb + (size_t)d * Npad + key);
      if (key + EPC > N) {                // last tile: zero the padded keys (0 * garbage must stay 0)
        T* e = reinterpret_cast<T*>(&u);
#pragma unroll
        for (int x = 0; x < EPC; ++x)
          if (key + x >= N) e[x] = from_f32<T>(0.0f);
      }
      rv[i] = u;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < K_CHUNKS; ++i) {
      const int c = tid + i * SD_NT;
      const int row = c / (HD / EPC), ch = c % (HD / EPC);
      *reinterpret_cast<uint4*>(Ks + row * Cfg::K_PITCH + ch * 16) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < V_CHUNKS; ++i) {
      const int c = tid + i * SD_NT;
      const int d = c / (SD_KB / EPC), ch = c % (SD_KB / EPC);
      char* dst = Vs + d * Cfg::V_PITCH + ch * 16;
      if (sizeof(T) == 2) {               // pitch 136 is only 8-byte aligned
        *reinterpret_cast<uint2*>(dst) = make_uint2(rv[i].x, rv[i].y);
        *reinterpret_cast<uint2*>(dst + 8) = make_uint2(rv[i].z, rv[i].w);
      } else {
        *reinterpret_cast<uint4*>(dst) = rv[i];
      }
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.0f; oacc[1][r] = 0.0f; }
  float m_run = -INFINITY, l_part = 0.0f;
  const float c2 = 0.125f * 1.44269504088896340736f;   // d^-0.5 * log2(e)

  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);

    // ---- S^T = K . Q^T : two 32-key blocks ----
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<T> fk;
        fk.load16B(reinterpret_cast<const T*>(Ks + (kb * 32 + li) * Cfg::K_PITCH) + ks * 16 + half * 8);
        sacc[kb] = mma32(fk, fq[ks], sacc[kb]);
      }
    }
    if (kt == nkt - 1 && (N % SD_KB) != 0) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * SD_KB + kb * 32 + acc_row(r, half) >= N) sacc[kb][r] = -INFINITY;
    }

    // ---- online softmax (this lane: one query, 32 of the 64 keys; partner lane^32 has the rest) ----
    float mloc = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[kb][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
    const float mc = m_new * c2;
    m_run = m_new;
    float psum = 0.0f;
    Frag<T> fp[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], c2, -mc));
        psum += p;
        fp[kb][r >> 3].set(r & 7, p);
      }
    l_part = l_part * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const char* vrow = Vs + (db * 32 + li) * Cfg::V_PITCH;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          Frag<T> fv;
          load_vt_frag(fv, vrow, kb * 32 + s * 16 + half * 4);
          oacc[db] = mma32(fv, fp[kb][s], oacc[db]);
        }
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      lstore();
      __syncthreads();
    }
  }

  // ---- epilogue: normalise, store O (heads concatenated) and lse ----
  const float l = l_part + __shfl_xor(l_part, 32);
  const float inv = 1.0f / l;
  if (query < N) {
    T* orow = o + ((size_t)b * N + query) * ((size_t)h * HD) + head * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * half;
        store4(orow + d, oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv, oacc[db][4 * g + 2] * inv,
               oacc[db][4 * g + 3] * inv);
      }
    if (half == 0) lse[(size_t)bh * N + query] = m_run * 0.125f + logf(l);
  }
}

template <typename T>
int launch_sdpa(const void* q, const void* k, const void* vt, void* o, float* lse, int B, int N, int h,
                hipStream_t s) {
  const int Npad = as_round_up(N, 64);
  const int grid = as_ceil_div(N, SD_QB) * B * h;
  const size_t lds = (size_t)SD_KB * SdpaCfg<T>::K_PITCH + (size_t)HD * SdpaCfg<T>::V_PITCH;
  hipLaunchKernelGGL((sdpa_fwd_kernel<T>), dim3(grid), dim3(SD_NT), lds, s, (const T*)q, (const T*)k,
                     (const T*)vt, (T*)o, lse, B, N, Npad, h);
  AS_CHECK_LAUNCH("sdpa_fwd");
  return AS_OK;
}

}  // namespace

extern "C" int as_sdpa_fwd(const void* q, const void* k, const void* vt, void* o, float* lse, int B, int N, int h,
                           int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && vt && o && lse, AS_E_BADARG, "as_sdpa_fwd: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && h > 0, AS_E_BADARG, "as_sdpa_fwd: bad sizes B=%d N=%d h=%d", B, N, h);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_sdpa<__bf16>(q, k, vt, o, lse, B, N, h, s);
  if (dtype == AS_F32) return launch_sdpa<float>(q, k, vt, o, lse, B, N, h, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_sdpa_fwd: dtype %d", dtype);
}

extern "C" int as_attn_fwd(const void* x, const void* Wqkv, const float* bqkv, const void* Wproj,
                           const float* bproj, void* out, float* lse, void* q, void* k, void* vt, void* o, int B,
                           int N, int D, int h, int dtype, as_stream_t stream) {
  int rc = as_qkv_fwd(x, Wqkv, bqkv, q, k, vt, B, N, D, h, dtype, stream);
  if (rc != AS_OK) return rc;
  rc = as_sdpa_fwd(q, k, vt, o, lse, B, N, h, dtype, stream);
  if (rc != AS_OK) return rc;
  return as_linear_fwd(o, Wproj, bproj, out, B * N, D, D, dtype, 0, stream);
}
