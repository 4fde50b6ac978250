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
#include <stdlib.h>
#include <utility>
#include "common.h"

namespace {

constexpr int SD_NT = 256, SD_QB = 128, SD_KB = 64, HD = 64;
constexpr int SD_REC = 68;                       // floats per row of a key-split partial record

// LDS row pitches of the generic (register-staged) kernel; only its fp32 instantiation is built -- bf16 always takes
// the LDS-DMA ring kernel below
template <typename T> struct SdpaCfg {
  static constexpr int K_PITCH = HD * (int)sizeof(T) + 16;      // bytes
  static constexpr int V_PITCH = SD_KB * (int)sizeof(T) + 16;
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
  constexpr int BUF_BYTES = SD_KB * Cfg::K_PITCH + HD * Cfg::V_PITCH;     // one K tile + one V^T tile

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
  auto lstore = [&](int buf) {
    char* Ks = smem + buf * BUF_BYTES;
    char* Vs = Ks + SD_KB * Cfg::K_PITCH;
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
  // q is stored pre-scaled by log2(e) / 8 (common.h): the MFMA scores ARE the base-2 logits

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);
    const char* Ks = smem + (kt & 1) * BUF_BYTES;
    const char* Vs = Ks + SD_KB * Cfg::K_PITCH;

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
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    const float mc = m_new;
    m_run = m_new;
    float psum = 0.0f;
    Frag<T> fp[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(sacc[kb][r] - mc);
        psum += p;
        fp[kb][r >> 3].set(r & 7, p);
      }
    l_part = l_part * alpha + psum;
    if (!__all(alpha == 1.0f)) {          // the running max moved for some query of this wave: rescale O
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
    }

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
    // the other buffer was last read in iteration kt-1, and every wave has passed the barrier that ended it
    if (kt + 1 < nkt) lstore((kt + 1) & 1);
    __syncthreads();
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
    if (half == 0) lse[(size_t)bh * N + query] = m_run * AS_LN2 + logf(l);
  }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 fast path.  Same dataflow as sdpa_fwd_kernel, but the K and V^T tiles go global -> LDS with
// global_load_lds_dwordx4 (no staging VGPRs) into a 3-deep LDS ring: tile kt+2 is in flight while tile kt is
// consumed, with counted `s_waitcnt vmcnt(N)` + raw s_barrier so the LDS-DMA spans the barriers.  The LDS images are
// lane-linear, so the bank swizzle sits on the source address: K row r (V^T row d) keeps global 16-B chunk c at
// c ^ ((r >> 1) & 7).  Rows are 128 B, so rows 2j and 2j+1 fill one 256-B bank row and share a key; the 16-lane
// groups ds_read_b128 is serviced in ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS) then hit 16 distinct slots.  Padded keys of the last tile are zeroed in LDS (P is exactly 0 there, but 0 * garbage must stay 0).
// ---------------------------------------------------------------------------------------------------------
// LDS reads the compiler must NOT see: after an LDS-DMA hipcc drains vmcnt(0) before any ds_read it can see (it cannot
// prove the read does not alias the DMA destination), which would collapse the 3-deep ring to depth 0.  The reads are
// issued in inline asm and completed by lds_wait<>, which names every destination as read-write so nothing is
// consumed early (cdna_hip_programming.md section 5.7, form ii), followed by a sched_barrier (rule 18).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void lds_read128(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
__device__ __forceinline__ void lds_read64(u32x2& dst, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr));
}
// immediate-offset forms: the per-lane address register is loop-invariant, ring slot / row block are immediates
template <int OFF> __device__ __forceinline__ void lds_read128_i(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_read64_i(u32x2& dst, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int... I, typename F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
__device__ __forceinline__ void lds_wait4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void lds_wait8(u32x2 (&v)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
  __builtin_amdgcn_sched_barrier(0);
}

#ifndef AS_SDPA_NBUF2
#define AS_SDPA_NBUF2 3           // LDS ring depth of sdpa_fwd_pipe_kernel<2, ...> (3 or 4 slots of 16 KiB)
#endif
#ifndef AS_SDPA_PRIO
#define AS_SDPA_PRIO 0            // experiments with static s_setprio (tools/experiments/sdpa_impl_bench.py)
#endif
#ifndef AS_SDPA_SGB
#ifndef AS_SDPA_LATE_QPIN
#define AS_SDPA_LATE_QPIN 1       // wait for the Q fragments behind the first K / V^T tiles' DMA, not in front of it
#endif
#ifndef AS_SDPA_WIDE_STORE
#define AS_SDPA_WIDE_STORE 1     // 16-byte O stores through v_permlane32_swap pairs (0: 8-byte stores per half-wave)
#endif
#define AS_SDPA_SGB 1             // sched_group_barrier interleave of sdpa_fwd_pipe_kernel's reference-free step
#endif
#ifndef AS_SDPA_NO_DEAD_SKIP
#define AS_SDPA_NO_DEAD_SKIP 0    // 1: waves without a valid query row run the full pass (A/B of the round-4 skip)
#endif
#ifndef AS_SDPA_DOT2SUM
#define AS_SDPA_DOT2SUM 0         // experiment: softmax row sums by v_dot2_f32_bf16 on the packed weights (8 per unit) instead of 16 adds
#endif
#ifndef AS_SDPA_ABLATE
#define AS_SDPA_ABLATE 0          // timing experiments only (tools/experiments/sdpa_ablate.py): 1 no exp2, 2 no softmax
#endif                            // VALU, 3 no P.V MFMAs, 4 no Q.K MFMAs, 5 no LDS-DMA in the loop, 6 no barrier (old
                                  // kernel); 11 .. 17: the same for sdpa_fwd_pipe_kernel's reference-free pass (17: no LDS
                                  // fragment reads in the loop)
constexpr int GL_TILE = SD_KB * HD * 2;          // 8 KiB per K (or V^T) tile
constexpr int GL_NBUF = 3;                       // LDS ring: tiles kt, kt+1, kt+2 (48 KiB per workgroup)

// SPLIT = false: one workgroup per (q-tile, image*head), whole key range, writes o and lse.
// SPLIT = true : q-tile `qt_fixed` only; workgroup (image*head, slice) takes key tiles [slice*per, (slice+1)*per) and
//                leaves its unnormalised partial (O^T fp32, reference max, row sum) in `part` for sdpa_combine_kernel.
template <bool SPLIT>
__global__ __launch_bounds__(SD_NT, 3) void sdpa_fwd_glds_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                                 const __bf16* __restrict__ vt, __bf16* __restrict__ o,
                                                                 float* __restrict__ lse, int B, int N, int Npad, int h,
                                                                 int qt_fixed, int nslices, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][K tile | V^T tile]
  const int BH = B * h;
  const int bid = blockIdx.x;
  const int bh = bid % BH, qt = SPLIT ? qt_fixed : bid / BH;
  const int slice = SPLIT ? bid / BH : 0;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int query = qt * SD_QB + wave * 32 + li;
  const int qclamped = min(query, N - 1);
  const int nkt_all = Npad / SD_KB;
  const int per = SPLIT ? (nkt_all + nslices - 1) / nslices : nkt_all;
  const int kt_off = slice * per;                                   // first key tile of this workgroup
  const int nkt = min(nkt_all, kt_off + per) - kt_off;              // key tiles of this workgroup (>= 1 by construction)

  Frag<__bf16> fq[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fq[ks].load16B(q + qf_frag((size_t)bh, Npad, qclamped, ks, half));
  // consume Q here so hipcc waits for these ordinary loads BEFORE the ring starts; otherwise it re-waits vmcnt(0) at
  // their first use inside the loop on every iteration and drains the LDS-DMA ring
  asm volatile("; Q fragments landed" : "+v"(fq[0].v), "+v"(fq[1].v), "+v"(fq[2].v), "+v"(fq[3].v));

  // loader: per tile each wave moves 2 one-KiB pieces of K and 2 of V^T (8 rows x 128 B each)
  const int lr = lane >> 3, lc = lane & 7;
  const char* srcK[2];
  const char* srcV[2];
  {
    const int r = wave * 8 + lr;                                   // key row of the K tile / d row of the V^T tile
    const int key = (r >> 1) & 7;
    srcK[0] = reinterpret_cast<const char*>(k + ((size_t)bh * Npad + r) * HD) + ((lc ^ key) << 4);
    srcV[0] = reinterpret_cast<const char*>(vt + ((size_t)bh * HD + r) * Npad) + ((lc ^ key) << 4);
    srcK[1] = srcK[0] + 32 * HD * 2;                               // rows r + 32: same swizzle key
    srcV[1] = srcV[0] + (size_t)32 * Npad * 2;
  }
  // running source pointers: tiles are staged strictly in order, so each call just advances them by one tile
  const char* pK[2] = {srcK[0] + (size_t)kt_off * SD_KB * HD * 2, srcK[1] + (size_t)kt_off * SD_KB * HD * 2};
  const char* pV[2] = {srcV[0] + (size_t)kt_off * SD_KB * 2, srcV[1] + (size_t)kt_off * SD_KB * 2};
  auto stage = [&](int /*kt*/, int buf) {
    char* base = smem + buf * (2 * GL_TILE);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = (wave + 4 * j) * 1024;                     // rows 8*wave.. and 32 + 8*wave..
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pK[j],
                                       (__attribute__((address_space(3))) void*)(base + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pV[j],
                                       (__attribute__((address_space(3))) void*)(base + GL_TILE + piece), 16, 0, 0);
      pK[j] += SD_KB * HD * 2;
      pV[j] += SD_KB * 2;
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.0f; oacc[1][r] = 0.0f; }
  float m_run = 0.0f, l_part = 0.0f;                 // m_run in base-2 logit units (q is pre-scaled, common.h)

  // Ring protocol: at the top of iteration kt tiles kt and kt+1 are in flight or landed; tile kt+2 is issued into
  // the buffer last read in iteration kt-1 (every wave passed the barrier that ended it).  Each wave issues 4 LDS-DMA
  // pieces per tile, so `vmcnt(8)` = "everything except the two newest tiles has landed" = tile kt is in LDS; the
  // barrier then publishes all waves' pieces.  Raw s_barrier: __syncthreads() would drain vmcnt(0).
  // Ring protocol (GL_NBUF = 3, ONE barrier per tile): entering iteration kt, tile kt has landed and been published
  // and tile kt+1 is in flight; tile kt+2 is issued into the buffer last read in iteration kt-1 (every wave passed the
  // barrier that ended it).  Each wave issues 4 LDS-DMA pieces per tile, so at the end of the iteration `vmcnt(4)`
  // = "everything but the newest tile has landed" = tile kt+1 is in LDS; the barrier publishes it and at the same
  // time certifies that tile kt is fully consumed.  Raw s_barrier: __syncthreads() would drain vmcnt(0).
  stage(0, 0);
  if (nkt > 1) {
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  // per-lane LDS addresses of the operand fragments are loop-invariant (12 registers); with the tile loop unrolled by
  // the ring depth the slot base and the 32-row block are immediates, so no address arithmetic is left in the loop
  const unsigned smem_base = lds_addr(smem);
  // Key permutation: MFMA row i of the S^T tile is fed K row pi(i) = i with bits 2 and 3 swapped.  Accumulator
  // registers 8*s2 .. 8*s2+7 of a lane in half `half` then hold the 8 CONSECUTIVE keys 16*s2 + 8*half + 0..7, which is
  // exactly the k-fragment of the P^T operand of the second MFMA, so P goes accumulator -> operand with no shuffle and
  // the V^T operand is one plain 16-byte fragment read.
  unsigned koff[4], voff[4];
  {
    const int krow = (li & 0x13) | ((li & 4) << 1) | ((li & 8) >> 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = smem_base + krow * 128 + ((((ks << 1) | half) ^ ((krow >> 1) & 7)) << 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) voff[c] = smem_base + li * 128 + ((((c << 1) | half) ^ ((li >> 1) & 7)) << 4);
  }
  float mc = 0.0f;
  for (int kt0 = 0; kt0 < nkt; kt0 += GL_NBUF) {
   static_for<GL_NBUF>([&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    const int kt = kt0 + slot;
    if (kt >= nkt) return;
    if (kt + 2 < nkt && AS_SDPA_ABLATE != 5) stage(kt + 2, (slot + 2) % GL_NBUF);
    char* Ks = smem + slot * (2 * GL_TILE);
    char* Vs = Ks + GL_TILE;
    const int ktg = kt_off + kt;                    // absolute key tile
    const bool ragged = (ktg == nkt_all - 1) && (N % SD_KB) != 0;
    if (ragged) {                                   // zero V^T columns of the padded keys (workgroup-uniform branch)
      for (int e = tid; e < HD * SD_KB; e += SD_NT) {
        const int d = e >> 6, key = e & 63;
        if (ktg * SD_KB + key >= N)
          *reinterpret_cast<__bf16*>(Vs + d * 128 + (((key >> 3) ^ ((d >> 1) & 7)) << 4) + (key & 7) * 2) = (__bf16)0.0f;
      }
      __syncthreads();
    }

    f32x16 sacc[2];
    static_for<2>([&](auto kb_c) {
      constexpr int kb = decltype(kb_c)::value;
      constexpr int OFF = slot * (2 * GL_TILE) + kb * 32 * 128;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.0f;
      u32x4 kf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) lds_read128_i<OFF>(kf[ks], koff[ks]);
      lds_wait4(kf[0], kf[1], kf[2], kf[3]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<__bf16> fk;
        fk.v = *reinterpret_cast<bf16x8*>(&kf[ks]);
        if (AS_SDPA_ABLATE == 4) { asm volatile("" :: "v"(fk.v)); sacc[kb][ks] += 1e-3f; }
        else sacc[kb] = mma32(fk, fq[ks], sacc[kb]);
      }
    });
    if (ragged) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (ktg * SD_KB + kb * 32 + 16 * (r >> 3) + 8 * half + (r & 7) >= N) sacc[kb][r] = -INFINITY;
    }

    // Softmax reference point: the row max of the FIRST tile only.  Any fixed reference gives the same softmax; the
    // running max exists to keep exp() in range, and fp32/bf16 have 2^127 of headroom, so the max (16 v_max3 + a
    // shuffle per tile) is only recomputed when a row sum leaves the safe range (> 1e20, inf or NaN): then the tile is
    // redone against the true max and O / l are rescaled exactly as in the usual online softmax.
    auto rowmax = [&]() {
      float m = sacc[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[kb][r]);
      return fmaxf(m, __shfl_xor(m, 32));
    };
    if (kt == 0) {
      m_run = rowmax();
      mc = m_run;
    }
    // four independent partial sums: one 32-deep chain of dependent v_add would sit on the critical path of the tile
    float ps4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    Frag<__bf16> fp[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#if AS_SDPA_ABLATE == 1
        const float p = (sacc[kb][r] - mc) * 1e-3f;
        ps4[r & 3] += p;
        fp[kb][r >> 3].set(r & 7, p);
#elif AS_SDPA_ABLATE == 2
        asm volatile("" :: "v"(sacc[kb][r]));
        fp[kb][r >> 3].set(r & 7, 0.001f);
#else
        const float p = __builtin_amdgcn_exp2f(sacc[kb][r] - mc);
        ps4[r & 3] += p;
        fp[kb][r >> 3].set(r & 7, p);
#endif
      }
    float psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
    if (__any(!(psum < 1e20f))) {                       // rare: re-reference this wave's rows to the true running max
      const float m_cand = fmaxf(m_run, rowmax());
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
      m_run = m_cand;
      mc = m_cand;
      l_part *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
      psum = 0.0f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(sacc[kb][r] - mc);
          psum += p;
          fp[kb][r >> 3].set(r & 7, p);
        }
    }
    l_part += psum;

    static_for<2>([&](auto db_c) {
      constexpr int db = decltype(db_c)::value;
      constexpr int OFF = slot * (2 * GL_TILE) + GL_TILE + db * 32 * 128;
      u32x4 vf[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) lds_read128_i<OFF>(vf[c], voff[c]);
      lds_wait4(vf[0], vf[1], vf[2], vf[3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        Frag<__bf16> fv;
        fv.v = *reinterpret_cast<bf16x8*>(&vf[c]);
        if (AS_SDPA_ABLATE == 3) { asm volatile("" :: "v"(fv.v), "v"(fp[c >> 1][c & 1].v)); oacc[db][c] += 1e-3f; }
        else oacc[db] = mma32(fv, fp[c >> 1][c & 1], oacc[db]);
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS reads of this tile are done
    if (kt + 2 < nkt) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // my pieces of tile kt+1 have landed (kt+2 may still fly)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (AS_SDPA_ABLATE != 6) __builtin_amdgcn_s_barrier();
   });
  }

  const float l = l_part + __shfl_xor(l_part, 32);
  if (SPLIT) {
    // partial record of row (wave*32 + li): [64 x O^T unnormalised | m | l | pad], SD_REC floats (16-B aligned rows);
    // rows past N are never combined
    float* rec = part + (((size_t)bh * nslices + slice) * SD_QB + wave * 32 + li) * SD_REC;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(rec + db * 32 + 8 * g + 4 * half) =
            make_float4(oacc[db][4 * g], oacc[db][4 * g + 1], oacc[db][4 * g + 2], oacc[db][4 * g + 3]);
    if (half == 0) { rec[64] = m_run; rec[65] = l; }
    return;
  }
  const float inv = 1.0f / l;
  if (query < N) {
    __bf16* orow = o + ((size_t)b * N + query) * ((size_t)h * HD) + head * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * half;
        store4(orow + d, oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv, oacc[db][4 * g + 2] * inv,
               oacc[db][4 * g + 3] * inv);
      }
    if (half == 0) lse[(size_t)bh * N + query] = m_run * AS_LN2 + logf(l);
  }
}

// ---------------------------------------------------------------------------------------------------------
// sdpa_fwd_pipe_kernel: the same dataflow (swapped MFMAs, pi key order, LDS-DMA ring, first-tile softmax reference) with
// the tile loop SOFTWARE-PIPELINED INSIDE EACH WAVE.  sdpa_fwd_glds_kernel runs a tile as four serial phases (K reads ->
// 8 MFMAs -> ~110 VALU -> V reads -> 8 MFMAs) and relies on the other waves of the SIMD to fill the pipes; the ablation
//  showed that they do not: every phase costs 18-25 % and none hides under another.  Here the work
// is cut into UNITS of 32 queries x 32 keys (4 QK MFMAs, 16 scores per lane, 4 PV MFMAs) and one STEP of the loop is
//     matrix pipe :  S(u+1) = K(u+1) . Q^T      (4 MFMAs)      then      O += V^T(u-1) . P(u-1)     (4 MFMAs)
//     VALU        :  P(u) = exp2(S(u) * c - m), row sums, bf16 packing                       (independent of both)
//     LDS         :  V^T fragments of unit u-1 at the top, K fragments of unit u+2 in the middle (half a step ahead)
// so every MFMA has ~7 independent VALU instructions of another unit to issue under it, in ONE wave.  NQ = 1: a wave
// owns 32 queries and a tile is two units (key halves), 3 workgroups per CU as before (same grid, same split tail).
// NQ = 2: a wave owns 64 queries (two query blocks), a tile is four units ordered (key half, query block) so each K / V^T
// fragment read feeds two MFMAs; 2 workgroups per CU.
//
// The LDS-DMA is issued from inline asm, so hipcc sees no vm-counted LDS write and treats the fragment reads as ordinary
// LDS loads: it places them, counts their lgkmcnt and interleaves them itself (the other kernel has to hide every
// ds_read in asm).  Ring protocol, ONE barrier per tile: in the middle of the first step of tile kt (after the last LDS
// read of tile kt-1, before the first of tile kt+1) every wave drains its reads, waits for its own pieces of tile kt+1
// (the only DMA outstanding), passes the barrier and issues tile kt+2 into the slot of tile kt-1.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_dma16(unsigned voff, const char* sbase, unsigned lds_dst) {
  // global_load_lds_dwordx4, saddr form: 64 lanes x 16 B from sbase + voff[lane] -> LDS [m0 + lane * 16]; m0 is saved
  // and restored in the same statement (cdna_hip_programming.md 5.7).  `sbase` and `lds_dst` must be wave-uniform.
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// MODE 0: softmax referenced to the row max of a query block's first unit, re-referenced when a unit sum leaves 1e20
//         (exactly sdpa_fwd_glds_kernel's arithmetic).
// MODE 1: reference-free first pass -- q is pre-scaled, so the MFMA output IS the base-2 logit and P = exp2(S) needs no
//         per-score subtraction: 16 exp2 + 16 adds + 8 packs per unit.  Any fixed reference gives the same softmax as
//         long as nothing leaves the fp32 range; a row whose sum ends outside [1e-30, 1e30] (logits beyond +-87 in
//         natural units: not something LayerNorm-ed ViT tokens produce, but legal input) makes the WORKGROUP run the
//         MODE 0 pass over its keys again.  Tested with spiked rows (tests/test_gpu_kernels.py).
// SPLIT 0: one workgroup per (q-tile, image-head), whole key range.  SPLIT 1: the key-split last q-tile (partials merged by
// sdpa_combine_kernel).  SPLIT 2 = STREAM-K (round 4): a persistent grid of G = 2 x #CU workgroups; the flattened
// (unit = 256-row q-tile of an image-head, key tile) space is cut into G equal contiguous ranges, so every resident slot
// carries the same number of tile steps -- at ViT-B / 1024^2 / B = 2 the plain grid is 408 workgroups on 512 slots (104 CUs
// host one workgroup and idle early), here 512 workgroups x 52.6 steps.  A unit cut by a range boundary is finished by
// the LAST of its pieces to arrive, inside the kernel: every other piece leaves (O unnormalised in bf16, m, l) in the
// workspace and the last arriver merges them in PIECE order with its own piece rounded to bf16 the same way, so the
// result does not depend on who arrives last (cdna_hip_programming.md: in-launch split-K reduction, counter form:
// agent-scope release by the writers, agent-scope acquire by the reducer, counters zeroed by a memset node per call).
constexpr int SK_REC_BYTES = 2 * SD_QB * HD * 2 + 2 * SD_QB * 8;  // one 256-row piece: bf16 O + float2 (m, l) per row
// NW = waves per workgroup (4; 8 = round-4 experiment AS_SDPA_IMPL=7: 512-row workgroups, one per CU, every K / V^T tile
// staged once for twice the queries -- the LDS-DMA is 14 % of the loop, profiles/r04_sdpa_streamk.md).
template <int NQ, int SPLIT, int MODE, int NW = 4>
__global__ __launch_bounds__(64 * NW, NQ == 1 ? 3 : 2) void sdpa_fwd_pipe_kernel(
    const __bf16* __restrict__ q, const __bf16* __restrict__ k, const __bf16* __restrict__ vt, __bf16* __restrict__ o,
    float* __restrict__ lse, int B, int N, int Npad, int h, int qt_fixed, int nslices, float* __restrict__ part,
    int mix_mode, int mix_a, int mix_r) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [3][K tile | V^T tile]
  constexpr bool KSPLIT = SPLIT == 1, SK = SPLIT == 2;
  static_assert(!SK || (NQ == 2 && NW == 4), "stream-K is built on the 256-row workgroup");
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int NT = 64 * NW;                                     // threads of a workgroup
  constexpr int PPW = 8 / NW;                                     // one-KiB pieces of K (and of V^T) a wave moves per tile
  constexpr int QROWS = 32 * NQ * NW;                             // query rows of a workgroup
  constexpr int UPT = 2 * NQ;                                     // units per tile
  constexpr int SLOTB = 2 * GL_TILE;                              // bytes of a ring slot
  constexpr int NBUF = NQ == 2 ? AS_SDPA_NBUF2 : 3;               // ring depth: NBUF - 2 tiles in flight beyond the next
  const int BH = B * h;
  const int bid = blockIdx.x;
  // Workgroup -> (image-head, first query row).  Plain grids: bid = tile * BH + bh.  MIXED launch (mix_mode 1 / 2, see
  // launch_sdpa_glds): the full 128-row chunks of every head are covered by 256-row workgroups of the NQ = 2 instance
  // (mode 1: head bh owns a_bh = mix_a + (bh < mix_r) of them, rows [0, 256 a_bh)) and by 128-row workgroups of the NQ = 1
  // instance (mode 2: the chunks after them); head-major numbering, closed form.
  int bh, row0;
  if (SPLIT != 0 || mix_mode == 0) {
    bh = bid % BH;
    row0 = (KSPLIT ? qt_fixed : bid / BH) * QROWS;              // (stream-K: set per segment below)
  } else {
    const int cf = N / SD_QB;                                     // full 128-row chunks of a head
    const int per_hi = mix_mode == 1 ? mix_a + 1 : cf - 2 * (mix_a + 1);   // tiles of the heads < mix_r / of the others
    const int per_lo = mix_mode == 1 ? mix_a : cf - 2 * mix_a;
    int t;
    if (bid < mix_r * per_hi) { bh = bid / per_hi; t = bid - bh * per_hi; }
    else { const int b2 = bid - mix_r * per_hi; bh = mix_r + b2 / per_lo; t = b2 % per_lo; }
    row0 = mix_mode == 1 ? t * (2 * SD_QB) : (mix_a + (bh < mix_r ? 1 : 0)) * (2 * SD_QB) + t * SD_QB;
  }
  const int slice = KSPLIT ? bid / BH : 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int nkt_all = Npad / SD_KB;
  const int per = KSPLIT ? (nkt_all + nslices - 1) / nslices : nkt_all;
  int kt_off = slice * per;                                       // (stream-K: the segment's key range, set below)
  int nkt = min(nkt_all, kt_off + per) - kt_off;                  // >= 1 by construction
  const bool has_ragged = (N % SD_KB) != 0;

  int query[NQ];
  bf16x8 fq[NQ][4];
  auto load_q = [&]() {
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      query[qb] = row0 + (wave * NQ + qb) * 32 + li;
      const int qc = min(query[qb], N - 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fq[qb][ks] = *reinterpret_cast<const bf16x8*>(q + qf_frag((size_t)bh, Npad, qc, ks, half));
    }
  };
  if (!SK) load_q();

  // loader: per tile each wave moves 2 one-KiB pieces of K and 2 of V^T (8 rows x 128 B each); the bank swizzle sits on
  // the source address.  Per-lane 32-bit offsets + a scalar tile base (advanced by one tile per stage() call).
  unsigned offK[2], offV[2];
  {
    const int lr = lane >> 3, lc = lane & 7;
    const int r = wave * 8 + lr;
    const int key = (r >> 1) & 7;
    offK[0] = r * (HD * 2) + ((lc ^ key) << 4);
    offK[1] = offK[0] + 32 * HD * 2;                               // (second piece: NW = 4 only)
    offV[0] = r * (Npad * 2) + ((lc ^ key) << 4);
    offV[1] = offV[0] + 32 * Npad * 2;
  }
  const char* k_first = reinterpret_cast<const char*>(k + ((size_t)bh * Npad + (size_t)kt_off * SD_KB) * HD);
  const char* v_first = reinterpret_cast<const char*>(vt + (size_t)bh * HD * Npad + (size_t)kt_off * SD_KB);
  const unsigned smem_base = lds_addr(smem);

  // per-lane fragment addresses (loop invariant): slot, key half and d block are immediates of the reads
  const char* kptr[4];
  const char* vptr[4];
  {
    const int krow = (li & 0x13) | ((li & 4) << 1) | ((li & 8) >> 1);                  // pi(li): bits 2 and 3 swapped
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kptr[ks] = smem + krow * 128 + ((((ks << 1) | half) ^ ((krow >> 1) & 7)) << 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) vptr[c] = smem + GL_TILE + li * 128 + ((((c << 1) | half) ^ ((li >> 1) & 7)) << 4);
  }
  // make hipcc wait for the Q fragments HERE (ordinary loads), not inside the loop
  auto pin_q = [&]() {
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) asm volatile("; Q fragments landed" : "+v"(fq[qb][0]), "+v"(fq[qb][1]), "+v"(fq[qb][2]), "+v"(fq[qb][3]));
  };
#if !AS_SDPA_LATE_QPIN
  if (!SK) pin_q();
#endif

  auto ring_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // zero the V^T columns of the padded keys of the LAST tile (P is exactly 0 there, but 0 * garbage must stay 0):
  // thread = (d row, 16-key group); followed by a barrier
  auto zero_pad_cols = [&](int slot) {
    const int nv = N - (nkt_all - 1) * SD_KB;                     // valid keys of the last tile, 1 .. 63
    char* Vs = smem + slot * SLOTB + GL_TILE;
    constexpr int TPR = NT / 64, CPT = 8 / TPR;                  // threads per d row, 16-byte chunks per thread
    const int d = tid / TPR;
#pragma unroll
    for (int cc = 0; cc < CPT; ++cc) {
      const int c = (tid % TPR) * CPT + cc;
      if (8 * c + 8 > nv) {
        bf16x8* pch = reinterpret_cast<bf16x8*>(Vs + d * 128 + ((c ^ ((d >> 1) & 7)) << 4));
        bf16x8 v = *pch;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (8 * c + t >= nv) v[t] = (__bf16)0.0f;
        *pch = v;
      }
    }
    ring_barrier();
  };
  auto load_k = [&](bf16x8 (&kf)[4], auto off_c) {                 // K fragments of one 32-key block: 4 x 16 B
    constexpr int OFF = decltype(off_c)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kptr[ks] + OFF);
  };
  auto rowmax16 = [&](const f32x16& s) {
    float m = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, s[r]);
    return fmaxf(m, __shfl_xor(m, 32));
  };
  auto mask_ragged = [&](f32x16& s, int kb) {                      // keys >= N of the last tile -> -inf
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((nkt_all - 1) * SD_KB + kb * 32 + 16 * (r >> 3) + 8 * half + (r & 7) >= N) s[r] = -INFINITY;
  };

  f32x16 oacc[NQ][2];
  float m_run[NQ], l_row[NQ];
#if AS_SDPA_PRIO == 1
  // experiment: static issue priority for every second round of workgroups (the second workgroup a CU receives)
  if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);
#elif AS_SDPA_PRIO == 2
  if (wave & 1) __builtin_amdgcn_s_setprio(1);
#elif AS_SDPA_PRIO == 3
  // MI355X_MICROARCH "static priority for the younger half": waves 4-7 of an 8-wave workgroup lose every arbitration by age
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#elif AS_SDPA_PRIO == 4
  if (NW == 8 && wave < 4) __builtin_amdgcn_s_setprio(1);
#endif

  // one pass over this workgroup's key tiles.  FASTP: reference-free (MODE 1 first pass)
  auto run_pass = [&](auto fast_c) {
    constexpr bool FASTP = decltype(fast_c)::value;
    const char* k_tile = k_first;                                  // scalar: next tile to stage
    const char* v_tile = v_first;
    auto stage = [&](int slot) {                                   // tiles are staged strictly in order
      const unsigned base = smem_base + slot * SLOTB;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const unsigned piece = (wave + NW * j) * 1024;
        lds_dma16(offK[j], k_tile, base + piece);
        lds_dma16(offV[j], v_tile, base + GL_TILE + piece);
      }
      k_tile += SD_KB * HD * 2;
      v_tile += SD_KB * 2;
    };
    stage(0);
    if (nkt > 1) stage(1);
    if (NBUF == 4 && nkt > 2) stage(2);
#if AS_SDPA_LATE_QPIN
    // the Q fragments (ordinary loads, issued at the top of the kernel) are waited for HERE, behind the first tiles' LDS-DMA:
    // the two round trips overlap instead of following each other (the wait belongs in front of the first consumer, T20
    // follow-on).  hipcc counts only its own loads, so its wait is vmcnt(0): it covers the tiles just staged as well -- they
    // were issued back to back and land together
    if (!SK) pin_q();
#endif
    // the slot "before tile 0" feeds the first step's (all-zero) P.V product: its V^T half must hold finite numbers
    {
      const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int o_ = 0; o_ < GL_TILE; o_ += NT * 16) *reinterpret_cast<uint4*>(smem + (NBUF - 1) * SLOTB + GL_TILE + o_ + tid * 16) = z;
    }
    // wait for tile 0 only: the later tiles' pieces (4 LDS-DMA per wave and tile) may stay in flight
    if (NBUF == 4 && nkt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPW) : "memory");
    else if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ring_barrier();
    if (has_ragged && kt_off == nkt_all - 1) zero_pad_cols(0);     // tile 0 of this workgroup is the ragged one

    float mc[NQ], lp4[NQ][4];
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      m_run[qb] = 0.0f; mc[qb] = 0.0f;
      lp4[qb][0] = lp4[qb][1] = lp4[qb][2] = lp4[qb][3] = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[qb][0][r] = 0.0f; oacc[qb][1][r] = 0.0f; }
    }
    // pipeline state: scores of the current unit, K fragments for the next one, P of the previous one
    f32x16 s_cur;
    bf16x8 kf[4], vf[4], p_prev[2];
#pragma unroll
    for (int t = 0; t < 8; ++t) {                                  // the first step's P.V product is 0 x 0
      p_prev[0][t] = (__bf16)0.0f; p_prev[1][t] = (__bf16)0.0f;
      vf[0][t] = (__bf16)0.0f; vf[1][t] = (__bf16)0.0f; vf[2][t] = (__bf16)0.0f; vf[3][t] = (__bf16)0.0f;
    }
    {
      const f32x16 zero = {0};
      load_k(kf, std::integral_constant<int, 0>{});
      s_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], fq[0][0], zero, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) s_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], fq[0][ks], s_cur, 0, 0, 0);
      if (NQ == 1) load_k(kf, std::integral_constant<int, 32 * 128>{});   // unit 1 = key half 1 (NQ = 2: same K, block 1)
      if (has_ragged && kt_off == nkt_all - 1) mask_ragged(s_cur, 0);
    }

    for (int kt0 = 0; kt0 < nkt; kt0 += NBUF) {
     static_for<NBUF>([&](auto slot_c) {
      constexpr int SLOT = decltype(slot_c)::value;
      const int kt = kt0 + SLOT;
      if (kt >= nkt) return;
      const int ktg = kt_off + kt;
      const bool ragged = has_ragged && ktg == nkt_all - 1;
      static_for<UPT>([&](auto i_c) {
        constexpr int I = decltype(i_c)::value;
        constexpr int QB = I % NQ;
        constexpr int IN = (I + 1) % UPT, QBN = IN % NQ;                                        // next unit
        constexpr int IP = (I + UPT - 1) % UPT, KBP = IP / NQ, QBP = IP % NQ;                    // previous unit
        constexpr int P_SLOT = I >= 1 ? SLOT : (SLOT + NBUF - 1) % NBUF;
        constexpr int INN = (I + 2) % UPT, KBNN = INN / NQ, QBNN = INN % NQ;                     // the unit after the next
        constexpr int NN_SLOT = I + 2 < UPT ? SLOT : (SLOT + 1) % NBUF;
        const f32x16 zero = {0};

        // (keys >= N of the ragged last tile were masked when these scores were produced: end of the previous step)
        // softmax reference of a query block = row max of its FIRST unit (see sdpa_fwd_glds_kernel)
        if (!FASTP && SLOT == 0 && I < NQ && kt == 0) {
          m_run[QB] = rowmax16(s_cur);
          mc[QB] = m_run[QB];
        }

        // ---- first half: V^T fragments of the previous unit; S(next) = K . Q^T; exp2 of this unit under the MFMAs ----
        if (QBP == 0 && AS_SDPA_ABLATE != 17) {
#pragma unroll
          for (int c = 0; c < 4; ++c)            // (d block c >> 1, 16-key step c & 1) of key half KBP
            vf[c] = *reinterpret_cast<const bf16x8*>(vptr[2 * KBP + (c & 1)] + P_SLOT * SLOTB + (c >> 1) * 32 * 128);
        }
        f32x16 s_next;
        if (AS_SDPA_ABLATE == 14) {                                // (timing experiment: no Q.K MFMAs)
          s_next = s_cur;
          asm volatile("" : "+v"(s_next) : "v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]));
        } else {
          s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], fq[QBN][0], zero, 0, 0, 0);
#pragma unroll
          for (int ks = 1; ks < 4; ++ks) s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], fq[QBN][ks], s_next, 0, 0, 0);
        }

        float p[16];
        float ps[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (FASTP) {
            if (AS_SDPA_ABLATE == 11) p[r] = s_cur[r] * 0.5f;       // (timing experiments: 11 no exp2, 12 no softmax VALU)
            else if (AS_SDPA_ABLATE == 12) p[r] = s_cur[r];
            else p[r] = __builtin_amdgcn_exp2f(s_cur[r]);
            if (AS_SDPA_ABLATE != 12 && !AS_SDPA_DOT2SUM) lp4[QB][r & 3] += p[r];
          } else {
            p[r] = __builtin_amdgcn_exp2f(s_cur[r] - mc[QB]);
            ps[r & 3] = r < 4 ? p[r] : ps[r & 3] + p[r];
          }
        }
        bf16x8 p_cur[2];
        if (AS_SDPA_ABLATE == 12 && FASTP) {                       // raw accumulator bits as "P": no VALU at all
          p_cur[0] = *reinterpret_cast<bf16x8*>(&p[0]);
          p_cur[1] = *reinterpret_cast<bf16x8*>(&p[8]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) p_cur[r >> 3][r & 7] = (__bf16)p[r];
          if (AS_SDPA_DOT2SUM && FASTP) {                          // experiment: row sums of the bf16-ROUNDED weights, 8 v_dot2 per unit
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
            for (int w_ = 0; w_ < 8; ++w_) {
              const bf16x2_t pr = {p_cur[w_ >> 2][2 * (w_ & 3)], p_cur[w_ >> 2][2 * (w_ & 3) + 1]};
              lp4[QB][w_ & 3] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, lp4[QB][w_ & 3], false);
            }
          }
        }

        if (I == 0) {
          // ---- ring hand-over (one barrier per tile) ----
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my reads of tile kt-1 are done
          if (kt + 1 < nkt) {
            // my pieces of tile kt+1 have landed (NBUF = 4: tile kt+2 may still be in flight)
            if (NBUF == 4 && kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (AS_SDPA_ABLATE != 16) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NBUF - 1 < nkt && AS_SDPA_ABLATE != 15) stage((SLOT + NBUF - 1) % NBUF);          // into the slot of tile kt-1
            if (has_ragged && ktg + 1 == nkt_all - 1) zero_pad_cols((SLOT + 1) % NBUF);
          }
        }

        // ---- second half: K fragments two units ahead; O += V^T . P of the previous unit ----
        if (QBNN == 0 && AS_SDPA_ABLATE != 17) load_k(kf, std::integral_constant<int, NN_SLOT * SLOTB + KBNN * 32 * 128>{});
        if (AS_SDPA_ABLATE == 13) {                                // (timing experiment: no P.V MFMAs)
          asm volatile("" : "+v"(oacc[QBP][0]), "+v"(oacc[QBP][1]) : "v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]), "v"(p_prev[0]), "v"(p_prev[1]));
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            oacc[QBP][c >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[c], p_prev[c & 1], oacc[QBP][c >> 1], 0, 0, 0);
        }

        if (!FASTP) {
          float psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
          if (__any(!(psum < 1e20f))) {                    // rare: re-reference this wave's rows to the true running max
            const float m_cand = fmaxf(m_run[QB], rowmax16(s_cur));
            const float alpha = __builtin_amdgcn_exp2f(m_run[QB] - m_cand);
            m_run[QB] = m_cand;
            mc[QB] = m_cand;
#pragma unroll
            for (int x = 0; x < 4; ++x) lp4[QB][x] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { oacc[QB][0][r] *= alpha; oacc[QB][1][r] *= alpha; }
            psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float pr = __builtin_amdgcn_exp2f(s_cur[r] - mc[QB]);
              psum += pr;
              p_cur[r >> 3][r & 7] = (__bf16)pr;
            }
          }
          lp4[QB][0] += psum;
        } else {
          // Issue order of the step (cdna_hip_programming.md T19): every MFMA is followed by 5 of the step's 40 VALU
          // (16 exp2, 16 adds, 8 packs) -- an in-order wave issues nothing while the next MFMA waits for the matrix pipe,
          // so back-to-back MFMAs leave their 32-cycle shadows empty and a clump of VALU leaves the pipe idle.  Left to
          // itself hipcc emits 16 exp2, then 8 MFMAs.  The V^T reads go under the first MFMA, the K reads under the fifth.
          // P(u) and the row sums must be COMPUTED in this step: hipcc's IR passes otherwise sink the exp2 / pack / add
          // chains down to their first use -- the P.V MFMAs of the NEXT step -- which undoes the software pipeline
          // (sched_barrier only binds the machine scheduler).  An empty volatile asm makes the values opaque here.
          asm volatile("" : "+v"(p_cur[0]), "+v"(p_cur[1]), "+v"(lp4[QB][0]), "+v"(lp4[QB][1]), "+v"(lp4[QB][2]), "+v"(lp4[QB][3]));
          if (AS_SDPA_SGB) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);   // VALU | TRANS
#pragma unroll
            for (int m = 1; m < 8; ++m) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (m == 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
            }
          }
          // without the rare-path branch a step is no longer its own basic block: keep hipcc from merging the steps'
          // schedules (it stretches live ranges across steps and spills)
          __builtin_amdgcn_sched_barrier(0);
        }
        // the scores of the NEXT unit: mask the padded keys if it lies in the ragged last tile.  Here, between two
        // steps, the (uniform, almost never taken) branch does not cut a step's schedule in two.
        {
          constexpr int KBN = IN / NQ;
          const bool ragged_next = I + 1 < UPT ? ragged : (has_ragged && ktg + 1 == nkt_all - 1);
          if (ragged_next) mask_ragged(s_next, KBN);
        }
        s_cur = s_next;
        p_prev[0] = p_cur[0];
        p_prev[1] = p_cur[1];
      });
     });
    }
    // drain: P.V of the last unit (its V^T fragments: key half 1 of the last tile)
    {
      constexpr int QBL = NQ - 1;
      const int last_slot = (nkt - 1) % NBUF;
      if (NQ == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          vf[c] = *reinterpret_cast<const bf16x8*>(vptr[2 + (c & 1)] + last_slot * SLOTB + (c >> 1) * 32 * 128);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        oacc[QBL][c >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[c], p_prev[c & 1], oacc[QBL][c >> 1], 0, 0, 0);
    }
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      const float lp = (lp4[qb][0] + lp4[qb][1]) + (lp4[qb][2] + lp4[qb][3]);
      l_row[qb] = lp + __shfl_xor(lp, 32);
    }
  };

  // A wave whose 64 query rows all lie beyond N (the last q-tile of an image-head: 101 valid rows of 256 at N = 4197 leave
  // waves 2 and 3 without a query) still owes the workgroup its share of the LDS-DMA and every barrier, but none of the
  // MFMA / softmax work: this is run_pass's ring protocol alone -- same staging order, same waits, same barriers.  Under
  // the chip's power limit (profiles/r04_power_probe.md) the work not done is clock for the waves that have rows.
  auto run_pass_dead = [&]() {
    const char* k_tile = k_first;
    const char* v_tile = v_first;
    auto stage = [&](int slot) {
      const unsigned base = smem_base + slot * SLOTB;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const unsigned piece = (wave + NW * j) * 1024;
        lds_dma16(offK[j], k_tile, base + piece);
        lds_dma16(offV[j], v_tile, base + GL_TILE + piece);
      }
      k_tile += SD_KB * HD * 2;
      v_tile += SD_KB * 2;
    };
    stage(0);
    if (nkt > 1) stage(1);
    if (NBUF == 4 && nkt > 2) stage(2);
    {
      const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int o_ = 0; o_ < GL_TILE; o_ += NT * 16) *reinterpret_cast<uint4*>(smem + (NBUF - 1) * SLOTB + GL_TILE + o_ + tid * 16) = z;
    }
    if (NBUF == 4 && nkt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPW) : "memory");
    else if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ring_barrier();
    if (has_ragged && kt_off == nkt_all - 1) zero_pad_cols(0);
    for (int kt = 0; kt < nkt; ++kt) {
      const int slot = kt % NBUF, ktg = kt_off + kt;
      if (kt + 1 < nkt) {
        if (NBUF == 4 && kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NBUF - 1 < nkt) stage((slot + NBUF - 1) % NBUF);
        if (has_ragged && ktg + 1 == nkt_all - 1) zero_pad_cols((slot + 1) % NBUF);
      }
    }
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      m_run[qb] = 0.0f;
      l_row[qb] = 1.0f;                                            // (never stored: every row of this wave is >= N)
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[qb][0][r] = 0.0f; oacc[qb][1][r] = 0.0f; }
    }
  };
  constexpr bool dead_skip_ok = NQ == 2 && !KSPLIT && AS_SDPA_ABLATE == 0;
  // the key range of the current (bh, row0, kt_off, nkt): reference-free pass, exact pass when a row sum left the range
  auto run_unit = [&]() {
    if (dead_skip_ok && row0 + wave * (NQ * 32) >= N && !AS_SDPA_NO_DEAD_SKIP) {
      run_pass_dead();
      if (MODE == 1) {                                             // the workgroup's vote on the exact pass: this wave abstains
        ring_barrier();
        int* flags = reinterpret_cast<int*>(smem);
        if (lane == 0) flags[wave] = 0;
        ring_barrier();
        int any_bad = 0;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) any_bad |= flags[w_];
        const bool redo = any_bad != 0;
        ring_barrier();
        if (redo) run_pass_dead();
      }
      return;
    }
    if (MODE == 1) {
      run_pass(std::true_type{});
      bool bad = false;
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) bad = bad || !(l_row[qb] > 1e-30f && l_row[qb] < 1e30f);
      // workgroup vote through the (now idle) ring; no static __shared__ object (it would shift the dynamic base)
      const int wave_bad = __any(bad) ? 1 : 0;              // (all lanes vote: not inside the lane-0 branch)
      ring_barrier();
      int* flags = reinterpret_cast<int*>(smem);
      if (lane == 0) flags[wave] = wave_bad;
      ring_barrier();
      int any_bad = 0;
#pragma unroll
      for (int w_ = 0; w_ < NW; ++w_) any_bad |= flags[w_];
      const bool redo = any_bad != 0;
      ring_barrier();
      if (redo && AS_SDPA_ABLATE == 0) run_pass(std::false_type{});     // (ablated builds produce garbage sums: no redo)
    } else {
      run_pass(std::false_type{});
    }
  };
  // o / lse of this workgroup's rows from (oacc, m_run, l_row)
  auto store_rows = [&]() {
    const int b = bh / h, head = bh % h;
#pragma unroll
    for (int qb = 0; qb < NQ; ++qb) {
      const float l = l_row[qb];
      const float inv = 1.0f / l;
#if AS_SDPA_WIDE_STORE
      // A query's row is split across the half-waves (lane i: columns 8k .. 8k+3, lane i + 32: 8k+4 .. 8k+7 of column group k):
      // one v_permlane32_swap per dword and PAIR of groups leaves lanes 0-31 with the 16 contiguous bytes of group k and lanes
      // 32-63 with those of group k + 1 -- 8 16-byte stores per row instead of 16 8-byte ones, same bytes, same addresses
      // (cdna_hip_programming.md T21: the tail is store-ISSUE-bound).  Every lane takes part in the swaps; rows >= N skip the store.
      uint2 o2[8];
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x4 v = {(__bf16)(oacc[qb][db][4 * g] * inv), (__bf16)(oacc[qb][db][4 * g + 1] * inv),
                            (__bf16)(oacc[qb][db][4 * g + 2] * inv), (__bf16)(oacc[qb][db][4 * g + 3] * inv)};
          o2[db * 4 + g] = __builtin_bit_cast(uint2, v);
        }
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        auto rx = __builtin_amdgcn_permlane32_swap(o2[k].x, o2[k + 1].x, false, false);
        auto ry = __builtin_amdgcn_permlane32_swap(o2[k].y, o2[k + 1].y, false, false);
        o2[k].x = rx[0]; o2[k + 1].x = rx[1];
        o2[k].y = ry[0]; o2[k + 1].y = ry[1];
      }
      if (query[qb] < N) {
        __bf16* orow = o + ((size_t)b * N + query[qb]) * ((size_t)h * HD) + head * HD + 8 * half;
#pragma unroll
        for (int k = 0; k < 8; k += 2)
          *reinterpret_cast<uint4*>(orow + 8 * k) = make_uint4(o2[k].x, o2[k].y, o2[k + 1].x, o2[k + 1].y);
        if (half == 0) lse[(size_t)bh * N + query[qb]] = m_run[qb] * AS_LN2 + logf(l);
      }
#else
      if (query[qb] < N) {
        __bf16* orow = o + ((size_t)b * N + query[qb]) * ((size_t)h * HD) + head * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d = db * 32 + 8 * g + 4 * half;
            store4(orow + d, oacc[qb][db][4 * g] * inv, oacc[qb][db][4 * g + 1] * inv, oacc[qb][db][4 * g + 2] * inv,
                   oacc[qb][db][4 * g + 3] * inv);
          }
        if (half == 0) lse[(size_t)bh * N + query[qb]] = m_run[qb] * AS_LN2 + logf(l);
      }
#endif
    }
  };

  if constexpr (SK) {
    // ---- stream-K: this workgroup's range of the flattened (unit, key tile) space ----
    const int G = (int)gridDim.x;
    const int nx = ((G & 7) == 0 && !(mix_mode & 1)) ? 8 : 1;      // block b runs on XCD b % 8 (observed): consecutive RANKS
    const int rank = (bid % nx) * (G / nx) + bid / nx;            // share an XCD, i.e. the K / V^T of a few heads per L2
    const int QT = (N + QROWS - 1) / QROWS;
    const int units = BH * QT;
    const long long T = (long long)units * nkt_all;
    long long st = T * rank / G;
    const long long st_end = T * (rank + 1) / G;
    int* const arrive = reinterpret_cast<int*>(part);
    int* const done = arrive + units;
    char* const recs = reinterpret_cast<char*>(part) + (((size_t)units * 8 + 255) & ~(size_t)255);
    bool first_seg = true;
    while (st < st_end) {
      const int u = (int)(st / nkt_all);
      const int kt0 = (int)(st - (long long)u * nkt_all);
      const int n_t = (int)((st_end - st) < (long long)(nkt_all - kt0) ? (st_end - st) : (long long)(nkt_all - kt0));
      // the pieces of unit u: rank r covers steps [T r / G, T (r + 1) / G); rank_of(x) = ((x + 1) G - 1) / T
      const long long ub = (long long)u * nkt_all;
      const int r_first = (int)(((ub + 1) * G - 1) / T);
      const int r_last = (int)(((ub + nkt_all) * G - 1) / T);
      const int np = r_last - r_first + 1, pidx = rank - r_first;
      if (!first_seg) ring_barrier();                             // every wave has left the previous segment's ring
      first_seg = false;
      bh = u / QT;
      row0 = (u - bh * QT) * QROWS;
      kt_off = kt0;
      nkt = n_t;
      k_first = reinterpret_cast<const char*>(k + ((size_t)bh * Npad + (size_t)kt_off * SD_KB) * HD);
      v_first = reinterpret_cast<const char*>(vt + (size_t)bh * HD * Npad + (size_t)kt_off * SD_KB);
      load_q();
      pin_q();
      run_unit();
      st += n_t;
      if (np == 1 || (mix_mode & 2)) {                            // (bit 1: timing experiment without the exchange -- wrong rows)
        store_rows();
        continue;
      }
      // ---- ticket: who finishes unit u ----
      ring_barrier();
      int* const box = reinterpret_cast<int*>(smem);
      if (tid == 0) box[0] = __hip_atomic_fetch_add(&arrive[u], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ring_barrier();
      const int ticket = __builtin_amdgcn_readfirstlane(box[0]);
      // piece p of unit u is rank r_first + p: its FIRST segment when p > 0 (slot 0), its last when p == 0 (slot 1)
      auto rec_of = [&](int p) { return recs + ((size_t)(r_first + p) * 2 + (p > 0 ? 0 : 1)) * SK_REC_BYTES; };
      if (ticket < np - 1) {
        char* rec = rec_of(pidx);
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb) {
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              bf16x8 v;
#pragma unroll
              for (int t = 0; t < 8; ++t) v[t] = (__bf16)oacc[qb][db][8 * gp + t];
              *reinterpret_cast<bf16x8*>(rec + ((((wave * NQ + qb) * 2 + db) * 2 + gp) * 64 + lane) * 16) = v;
            }
          if (half == 0)
            *reinterpret_cast<float2*>(rec + QROWS * HD * 2 + ((wave * NQ + qb) * 32 + li) * 8) = make_float2(m_run[qb], l_row[qb]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler may drop the fence's own wait here)
          __hip_atomic_fetch_add(&done[u], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        continue;
      }
      // ---- last arriver: the other pieces have all drawn their tickets, i.e. they are running: a bounded wait ----
      if (tid == 0) {
        while (__hip_atomic_load(&done[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < np - 1) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      float mx[NQ];
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) mx[qb] = m_run[qb];
      for (int p = 0; p < np; ++p) {
        if (p == pidx) continue;
        const char* rec = rec_of(p);
#pragma unroll
        for (int qb = 0; qb < NQ; ++qb)
          mx[qb] = fmaxf(mx[qb], reinterpret_cast<const float2*>(rec + QROWS * HD * 2 + ((wave * NQ + qb) * 32 + li) * 8)->x);
      }
      f32x16 res[NQ][2];
      float lsum[NQ];
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) {
        lsum[qb] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { res[qb][0][r] = 0.0f; res[qb][1][r] = 0.0f; }
      }
      for (int p = 0; p < np; ++p) {                              // PIECE order, whoever holds which piece
        if (p == pidx) {
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb) {
            const float w = __builtin_amdgcn_exp2f(m_run[qb] - mx[qb]);
            lsum[qb] += w * l_row[qb];
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
              for (int r = 0; r < 16; ++r) res[qb][db][r] += w * (float)(__bf16)oacc[qb][db][r];
          }
        } else {
          const char* rec = rec_of(p);
          bf16x8 v[NQ][2][2];
          float2 ml[NQ];
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
              for (int gp = 0; gp < 2; ++gp)
                v[qb][db][gp] = *reinterpret_cast<const bf16x8*>(rec + ((((wave * NQ + qb) * 2 + db) * 2 + gp) * 64 + lane) * 16);
            ml[qb] = *reinterpret_cast<const float2*>(rec + QROWS * HD * 2 + ((wave * NQ + qb) * 32 + li) * 8);
          }
#pragma unroll
          for (int qb = 0; qb < NQ; ++qb) {
            const float w = __builtin_amdgcn_exp2f(ml[qb].x - mx[qb]);
            lsum[qb] += w * ml[qb].y;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
              for (int r = 0; r < 16; ++r) res[qb][db][r] += w * (float)v[qb][db][r >> 3][r & 7];
          }
        }
      }
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) {
        m_run[qb] = mx[qb];
        l_row[qb] = lsum[qb];
        oacc[qb][0] = res[qb][0];
        oacc[qb][1] = res[qb][1];
      }
      store_rows();
      if (tid == 0) {                                             // leave the counters as the next call needs them
        __hip_atomic_store(&arrive[u], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&done[u], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  } else {
    run_unit();
    if (KSPLIT) {
#pragma unroll
      for (int qb = 0; qb < NQ; ++qb) {
        float* rec = part + (((size_t)bh * nslices + slice) * QROWS + (wave * NQ + qb) * 32 + li) * SD_REC;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(rec + db * 32 + 8 * g + 4 * half) =
                make_float4(oacc[qb][db][4 * g], oacc[qb][db][4 * g + 1], oacc[qb][db][4 * g + 2], oacc[qb][db][4 * g + 3]);
        if (half == 0) { rec[64] = m_run[qb]; rec[65] = l_row[qb]; }
      }
    } else {
      store_rows();
    }
  }
}

// merges the key-slice partials of the split q-tile: thread = (row, 4 consecutive d); grid (8, B*h), 256 threads
__global__ __launch_bounds__(256) void sdpa_combine_kernel(const float* __restrict__ part, __bf16* __restrict__ o,
                                                           float* __restrict__ lse, int B, int N, int h, int qt,
                                                           int nslices) {
  const int bh = blockIdx.y, b = bh / h, head = bh % h;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), dg = threadIdx.x & 15;
  const int query = qt * SD_QB + row;
  if (query >= N) return;
  const float* base = part + ((size_t)bh * nslices * SD_QB + row) * SD_REC;
  const size_t step = (size_t)SD_QB * SD_REC;
  float m = -INFINITY;
  for (int s = 0; s < nslices; ++s) m = fmaxf(m, base[s * step + 64]);
  float l = 0.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  for (int s = 0; s < nslices; ++s) {
    const float* r = base + s * step;
    const float sc = __builtin_amdgcn_exp2f(r[64] - m);
    const float4 v = *reinterpret_cast<const float4*>(r + dg * 4);
    l += r[65] * sc;
    a0 += v.x * sc; a1 += v.y * sc; a2 += v.z * sc; a3 += v.w * sc;
  }
  const float inv = 1.0f / l;
  store4(o + ((size_t)b * N + query) * ((size_t)h * HD) + head * HD + dg * 4, a0 * inv, a1 * inv, a2 * inv, a3 * inv);
  if (dg == 0) lse[(size_t)bh * N + query] = m * AS_LN2 + logf(l);
}

// ---------------------------------------------------------------------------------------------------------
// The 792-vs-768 tail.  At ViT-B / 1024^2 / B=2 the plain grid is 33 q-tiles x 24 (image, head) = 792 workgroups for
// 768 resident slots (3 per CU): the 24 left-over workgroups run alone afterwards and cost ~20% of the launch
// (measured: 138 us for the first 768, 172 us in total).  When dropping the LAST q-tile of every (image, head) makes
// the main grid an exact multiple of the slots and the caller supplied a workspace, that q-tile is computed by the
// same kernel in SPLIT mode -- (image*head) x 11 key slices = 264 short workgroups that still share K/V tiles through
// the LDS ring -- followed by a tiny merge of the 11 partial (max, sum, O) records per row (fixed order).
// ---------------------------------------------------------------------------------------------------------
// Which forward kernel.  AS_SDPA_IMPL (read per call; tests and tools/experiments/sdpa_impl_bench.py) forces one:
// 0 = sdpa_fwd_glds_kernel; sdpa_fwd_pipe_kernel: 1 = <NQ 1, MODE 0>, 2 = <2, 0>, 3 = <1, 1>, 4 = <2, 1>, 6 = stream-K,
// 7 = <2, 1> on 8-wave / 512-row workgroups.  Unset: the
// reference-free pipelined kernel with the query blocking (3 or 4) that sdpa_pick() prices cheaper for the shape.
int sdpa_impl_forced() {
  const char* e = getenv("AS_SDPA_IMPL");
  return e ? atoi(e) : -1;
}

int sdpa_slots() {                                    // resident workgroups of sdpa_fwd_glds_kernel: 3 per CU
  static int slots = 0;                               // read-only device-properties cache
  if (slots == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    slots = 3 * (cus > 0 ? cus : 256);
  }
  return slots;
}

// key slices of a split q-tile: about one workgroup per CU, no empty slice
int sdpa_tail_slices(int B, int N, int h) {
  const int nkt = as_round_up(N, 64) / SD_KB;
  int ns = as_ceil_div(sdpa_slots() / 3, B * h);
  ns = ns < 2 ? 2 : (ns > nkt ? nkt : ns);
  return as_ceil_div(nkt, as_ceil_div(nkt, ns));
}

// MIXED launch (AS_SDPA_IMPL=5; an experiment kept behind the switch -- built to get 3 query blocks per SIMD instead of 4,
// measured SLOWER: 148 us against 122 at ViT-B / 1024^2 / B = 2, the dispatcher does not pair the two kernels' workgroups
// one of each per CU; sdpa_pick() never chooses it):
// X workgroups of 256 rows (NQ = 2) + Y of 128 rows (NQ = 1) that together cover the full 128-row chunks
// of every head, X, Y <= #CUs, so that every CU hosts ONE of each (a SIMD holds one 229-register and one 164-register
// wave; a second of either does not fit) = 3 query blocks per SIMD instead of the 4 of two 256-row workgroups; the
// ragged rows after the last full chunk go through the key-split kernel.  At ViT-B / 1024^2 / B = 2: 24 heads x 32 chunks
// = 768 = 2 * 256 + 256.  Returns false when the shape does not decompose that way.
struct SdpaMix { int X, Y, a, r, cf; };
bool sdpa_mix_plan(int B, int N, int h, SdpaMix* m) {
  const int BH = B * h, cus = sdpa_slots() / 3, cf = N / SD_QB, C = BH * cf;
  if (cf < 2) return false;
  int X = (C + 1) / 3;
  if (X > cus) X = cus;
  const int Y = C - 2 * X;
  const int a = X / BH, r = X % BH;
  if (Y < 0 || Y > cus || 2 * (a + (r > 0 ? 1 : 0)) > cf) return false;
  *m = SdpaMix{X, Y, a, r, cf};
  return true;
}

int sdpa_split_slices(int B, int N, int h) {           // 0 = no split for this shape
  const int BH = B * h, qtiles = as_ceil_div(N, SD_QB), slots = sdpa_slots();
  bool tail = qtiles * BH > slots && ((qtiles - 1) * BH) % slots == 0;
  if (const char* e = getenv("AS_SDPA_TAIL")) tail = e[0] == '1' ? true : (e[0] == '0' ? false : tail);
  if (!tail) return 0;
  const int nkt = as_round_up(N, 64) / SD_KB;
  int ns = as_ceil_div(slots / 3, BH);                 // about one workgroup per CU
  ns = ns < 2 ? 2 : (ns > nkt ? nkt : ns);
  const int per = as_ceil_div(nkt, ns);
  return as_ceil_div(nkt, per);                         // no empty slice
}

// NQ = 1 (128-row workgroups, 3 per CU, key-split tail) or NQ = 2 (256-row workgroups, 2 per CU, no tail)?  Per key
// tile a CU that hosts w concurrent workgroups needs (measured on MI355X, profiles/r03_sdpa_variants.md):
//   NQ = 1:  0.60 / 1.28 / 1.64 us for w = 1 / 2 / 3        NQ = 2:  1.15 us for w = 1, 1.75 .. 2.0 us for w = 2
// and the key-split tail of NQ = 1 costs ~20 us per 66 tiles.  At ViT-B / 1024^2 / B = 2 (24 x 4197 rows): NQ = 1 is 768
// workgroups + tail = 130 us, NQ = 2 is 408 workgroups on 512 slots = 121 us; at N = 4096 (no tail) NQ = 1 wins, 105 us.
// ---- stream-K (impl 6, sdpa_fwd_pipe_kernel<2, 2, 1>) ----
int sdpa_sk_grid_forced() {                                       // test hook: AS_SDPA_SK_GRID=<workgroups> (small shapes)
  const char* e = getenv("AS_SDPA_SK_GRID");
  const int g = e ? atoi(e) : 0;
  return g > 0 && g <= 4096 ? g : 0;
}
int sdpa_sk_grid() {                                              // one workgroup per resident slot (2 per CU)
  const int f = sdpa_sk_grid_forced();
  return f > 0 ? f : 2 * (sdpa_slots() / 3);
}
size_t sdpa_sk_counter_bytes(int B, int N, int h) {
  const size_t units = (size_t)B * h * as_ceil_div(N, 2 * SD_QB);
  return (units * 8 + 255) & ~(size_t)255;                        // arrive[units], done[units]
}
size_t sdpa_sk_ws_bytes(int B, int N, int h) {
  return sdpa_sk_counter_bytes(B, N, h) + (size_t)sdpa_sk_grid() * 2 * SK_REC_BYTES;
}
// worth it when the plain 256-row grid leaves a fractional round and every workgroup still gets a long range
bool sdpa_sk_ok(int B, int N, int h) {
  const long long units = (long long)B * h * as_ceil_div(N, 2 * SD_QB), tiles = as_round_up(N, 64) / SD_KB;
  const int G = sdpa_sk_grid();
  if (sdpa_sk_grid_forced() > 0) return units * tiles / G >= 2;   // (the kernel itself takes any number of pieces per unit)
  return G > 0 && units * tiles / G >= 16 && units * 2 >= G;      // >= 16 tile steps per workgroup, <= 3 pieces per unit
}

int sdpa_pick(int B, int N, int h, bool tail_ok, bool sk_ok = false) {
  const int BH = B * h, cus = sdpa_slots() / 3;
  const double tiles = (double)as_ceil_div(N, SD_KB);
  auto layers = [&](int wgs, int per_cu, const double* cost, double partial_hi) {   // us per key tile
    const int full = wgs / (per_cu * cus), rem = wgs % (per_cu * cus);
    double t = full * cost[per_cu - 1];
    if (rem > 0) {
      const int w = as_ceil_div(rem, cus);                      // concurrent workgroups on the busiest CUs
      double c = cost[w - 1];
      if (w == per_cu && partial_hi > 0) c = partial_hi + (cost[w - 1] - partial_hi) * (double)(rem - (w - 1) * cus) / cus;
      t += c;
    }
    return t;
  };
  const double c1[3] = {0.60, 1.28, 1.64}, c2[2] = {1.15, 2.0};
  int q1 = as_ceil_div(N, SD_QB);
  const bool tail = tail_ok && sdpa_split_slices(B, N, h) > 0;
  const double t1 = tiles * layers((tail ? q1 - 1 : q1) * BH, 3, c1, 0.0) + (tail ? 0.30 * tiles : 0.0);
  const double t2 = tiles * layers(as_ceil_div(N, 2 * SD_QB) * BH, 2, c2, 1.75);
  // Stream-K is NOT picked by default: measured on MI355X (profiles/r04_sdpa_streamk.md) the balanced grid does not beat
  // the plain one -- 119.7 us with the exchange switched off against 117.3 us, 143.8 us with it -- because the chip delivers
  // the same ~228 tile steps per us whether 152 or all 256 CUs carry two workgroups: the idle CUs of the plain grid are
  // not lost time, the busy ones clock higher.  AS_SDPA_SK=1 puts it back into the model (it is kept tested through
  // AS_SDPA_IMPL=6).
  static const bool sk_model = getenv("AS_SDPA_SK") != nullptr;
  if (sk_model && sk_ok && sdpa_sk_ok(B, N, h)) {
    // every slot runs units * tiles / G steps at the two-per-CU rate, + the measured ~24 us of exchange and ~5 of memset
    const double tsk = (double)as_ceil_div(N, 2 * SD_QB) * BH * tiles / sdpa_sk_grid() * c2[1] + 29.0;
    if (tsk < t1 && tsk < t2) return 6;
  }
  if (t2 < t1) {
    // 64 queries per wave: on 512-row workgroups of EIGHT waves (impl 7) when they still cover at least half the CUs -- each
    // K / V^T tile is then staged once for twice the queries (the LDS-DMA is 14 % of the loop); measured in one call
    // (tools/experiments/sdpa_impl_bench.py): config 2 117.4 -> 114.4 us, ViT-L 1 x 16 x 6501 176.3 -> 166.8 us (1038 TFLOP/s),
    // 2 x 16 x 4096 128.9 -> 126.8 us; bitwise the same rows as the 256-row grid.  AS_SDPA_NO_W8=1 keeps four waves.
    static const bool no_w8 = getenv("AS_SDPA_NO_W8") != nullptr;
    return (!no_w8 && as_ceil_div(N, 512) * BH >= cus / 2) ? 7 : 4;
  }
  return 3;
}

int launch_sdpa_glds(const void* q, const void* k, const void* vt, void* o, float* lse, void* ws, size_t ws_bytes, int B,
                     int N, int h, hipStream_t s) {
  const int Npad = as_round_up(N, 64);
  const int BH = B * h;
  int qtiles = as_ceil_div(N, SD_QB);
  int ns = sdpa_split_slices(B, N, h);
  if (ns > 0 && (ws == nullptr || ws_bytes < (size_t)BH * ns * SD_QB * SD_REC * sizeof(float))) ns = 0;
  const int forced = sdpa_impl_forced();
  const bool sk_ws = ws != nullptr && ws_bytes >= sdpa_sk_ws_bytes(B, N, h) && getenv("AS_SDPA_NO_SK") == nullptr;
  int impl = forced >= 0 ? forced : sdpa_pick(B, N, h, ns > 0, sk_ws);
  if (impl == 6 && !(sk_ws && sdpa_sk_ok(B, N, h))) impl = sdpa_pick(B, N, h, ns > 0);
  SdpaMix mix{};
  const int tail_rows = N % SD_QB;
  int ns_t = 0;
  bool mixed = impl == 5 && sdpa_mix_plan(B, N, h, &mix);
  if (mixed && tail_rows > 0) {
    ns_t = sdpa_tail_slices(B, N, h);
    if (ws == nullptr || ws_bytes < (size_t)BH * ns_t * SD_QB * SD_REC * sizeof(float)) mixed = false;
  }
  if (impl == 5 && !mixed) impl = sdpa_pick(B, N, h, ns > 0) == 5 ? 4 : sdpa_pick(B, N, h, ns > 0);
  if (impl == 2 || impl == 4 || impl == 6 || impl == 7) ns = 0;   // 64 queries per wave: no split tail
  if (ns > 0) --qtiles;
  const size_t lds = (size_t)GL_NBUF * 2 * GL_TILE;        // 48 KiB
  // The split q-tile runs CONCURRENTLY with the main grid on a helper stream (fork / join with events on the caller's
  // stream): its 264 short workgroups fill the slots the main grid's workgroups free up as they retire, instead of
  // running as a separate 13 us phase afterwards.  Helper stream and events are created once per host thread.
  struct Side {                                          // one set PER DEVICE (as AsSide in common.h): a thread that alternates
    enum { kMaxDev = 16 };                               // between devices reuses each device's set instead of leaking one per
    struct Slot {                                        // switch; a set whose creation fails half-way is destroyed and stays off
      hipStream_t st = nullptr, st2 = nullptr;
      hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr;
      int state = 0;                                     // 0 = not tried, 1 = ready, -1 = failed
    };
    Slot slots[kMaxDev];
    hipStream_t st = nullptr, st2 = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr;
    bool ok = false;
    static void drop(Slot& sl) {
      if (sl.join2) (void)hipEventDestroy(sl.join2);
      if (sl.join) (void)hipEventDestroy(sl.join);
      if (sl.fork) (void)hipEventDestroy(sl.fork);
      if (sl.st2) (void)hipStreamDestroy(sl.st2);
      if (sl.st) (void)hipStreamDestroy(sl.st);
      sl = Slot();
    }
    void init() {
      int d = -1;
      ok = false;
      if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) return;
      Slot& sl = slots[d];
      if (sl.state == 0) {
        const bool made = hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking) == hipSuccess &&
                          hipStreamCreateWithFlags(&sl.st2, hipStreamNonBlocking) == hipSuccess &&
                          hipEventCreateWithFlags(&sl.fork, hipEventDisableTiming) == hipSuccess &&
                          hipEventCreateWithFlags(&sl.join, hipEventDisableTiming) == hipSuccess &&
                          hipEventCreateWithFlags(&sl.join2, hipEventDisableTiming) == hipSuccess;
        if (!made) {
          drop(sl);
          (void)hipGetLastError();                       // the launch falls back to the caller's stream: not its error
        }
        sl.state = made ? 1 : -1;
      }
      if (sl.state != 1) return;
      st = sl.st; st2 = sl.st2; fork = sl.fork; join = sl.join; join2 = sl.join2;
      ok = true;
    }
    ~Side() {                                            // thread exit
      for (Slot& sl : slots)
        if (sl.state == 1) drop(sl);
      (void)hipGetLastError();
    }
  };
  static thread_local Side side;
  if (ns > 0 || mixed) side.init();
  const bool concurrent = ns > 0 && qtiles > 0 && side.ok && getenv("AS_SDPA_SERIAL") == nullptr;
  hipStream_t s2 = concurrent ? side.st : s;
  if (concurrent) {
    if (hipEventRecord(side.fork, s) != hipSuccess || hipStreamWaitEvent(side.st, side.fork, 0) != hipSuccess) s2 = s;
  }
#define AS_PIPE_LAUNCH_MIX(NQ_, SPLIT_, MODE_, GRID_, STREAM_, QT_, NS_, WS_, MIX_, MIXA_, MIXR_)                                                  \
  hipLaunchKernelGGL((sdpa_fwd_pipe_kernel<NQ_, SPLIT_, MODE_>), dim3(GRID_), dim3(SD_NT), lds, STREAM_, (const __bf16*)q,  \
                     (const __bf16*)k, (const __bf16*)vt, (__bf16*)o, lse, B, N, Npad, h, QT_, NS_, (float*)(WS_), MIX_, MIXA_, MIXR_)
#define AS_PIPE_LAUNCH(NQ_, SPLIT_, MODE_, GRID_, STREAM_, QT_, NS_, WS_) \
  AS_PIPE_LAUNCH_MIX(NQ_, SPLIT_, MODE_, GRID_, STREAM_, QT_, NS_, WS_, 0, 0, 0)
  if (mixed) {
    // three concurrent launches: 256-row workgroups on the caller's stream, 128-row workgroups and the key-split ragged
    // rows on two helper streams (fork / join with events), so the dispatcher can place one of each kind on every CU
    static std::atomic<bool> attr_mix{false};
    const size_t lds2 = (size_t)AS_SDPA_NBUF2 * 2 * GL_TILE;
    if (!attr_mix && lds2 > 64 * 1024 - 1) {
      (void)hipFuncSetAttribute((const void*)sdpa_fwd_pipe_kernel<2, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      attr_mix = true;
    }
    bool forked = side.ok && getenv("AS_SDPA_SERIAL") == nullptr && hipEventRecord(side.fork, s) == hipSuccess && hipStreamWaitEvent(side.st, side.fork, 0) == hipSuccess &&
                  hipStreamWaitEvent(side.st2, side.fork, 0) == hipSuccess;
    hipStream_t sa = forked ? side.st : s, sb = forked ? side.st2 : s;
    {
      const size_t lds = lds2;
      AS_PIPE_LAUNCH_MIX(2, false, 1, mix.X, s, 0, 1, nullptr, 1, mix.a, mix.r);
    }
    AS_CHECK_LAUNCH("sdpa_fwd_pipe<2> (mixed)");
    if (mix.Y > 0) {
      AS_PIPE_LAUNCH_MIX(1, false, 1, mix.Y, sa, 0, 1, nullptr, 2, mix.a, mix.r);
      AS_CHECK_LAUNCH("sdpa_fwd_pipe<1> (mixed)");
    }
    if (tail_rows > 0) {
      AS_PIPE_LAUNCH(1, true, 1, ns_t * BH, sb, mix.cf, ns_t, ws);
      AS_CHECK_LAUNCH("sdpa_fwd<split> (mixed)");
      hipLaunchKernelGGL(sdpa_combine_kernel, dim3(SD_QB / 16, BH), dim3(256), 0, sb, (const float*)ws, (__bf16*)o, lse, B, N,
                         h, mix.cf, ns_t);
      AS_CHECK_LAUNCH("sdpa_combine (mixed)");
    }
    if (forked) {
      (void)hipEventRecord(side.join, side.st);
      (void)hipEventRecord(side.join2, side.st2);
      (void)hipStreamWaitEvent(s, side.join, 0);
      (void)hipStreamWaitEvent(s, side.join2, 0);
    }
    return AS_OK;
  }
  if (impl == 7) {                                           // 512-row workgroups of 8 waves (64 queries per wave), one per CU
    const size_t lds = (size_t)AS_SDPA_NBUF2 * 2 * GL_TILE;
    static std::atomic<bool> attr_w8{false};
    if (!attr_w8 && lds > 64 * 1024 - 1) {
      (void)hipFuncSetAttribute((const void*)sdpa_fwd_pipe_kernel<2, 0, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_w8 = true;
    }
    hipLaunchKernelGGL((sdpa_fwd_pipe_kernel<2, 0, 1, 8>), dim3(as_ceil_div(N, 512) * BH), dim3(512), lds, s, (const __bf16*)q,
                       (const __bf16*)k, (const __bf16*)vt, (__bf16*)o, lse, B, N, Npad, h, 0, 1, (float*)nullptr, 0, 0, 0);
    AS_CHECK_LAUNCH("sdpa_fwd_pipe<8 waves>");
    return AS_OK;
  }
  if (impl == 6) {                                           // stream-K: persistent, balanced, merged in the kernel
    const size_t lds = (size_t)AS_SDPA_NBUF2 * 2 * GL_TILE;
    static std::atomic<bool> attr_sk{false};
    if (!attr_sk && lds > 64 * 1024 - 1) {
      (void)hipFuncSetAttribute((const void*)sdpa_fwd_pipe_kernel<2, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_sk = true;
    }
    const char* dbg_e = getenv("AS_SDPA_SK_DEBUG");               // experiments: 1 identity ranks, 2 no exchange, 4 no memset
    const int dbg = dbg_e ? atoi(dbg_e) : 0;
    if (!(dbg & 4) && hipMemsetAsync(ws, 0, sdpa_sk_counter_bytes(B, N, h), s) != hipSuccess) {
      as_set_error("as_sdpa_fwd: clearing the stream-K counters failed");
      return AS_E_LAUNCH;
    }
    AS_PIPE_LAUNCH_MIX(2, 2, 1, sdpa_sk_grid(), s, 0, 1, ws, dbg & 3, 0, 0);
    AS_CHECK_LAUNCH("sdpa_fwd_pipe<stream-K>");
    return AS_OK;
  }
  if (impl == 2 || impl == 4) {                              // 64 queries per wave, 256 per workgroup, no split tail
    const int grid2 = as_ceil_div(N, 2 * SD_QB) * BH;
    const size_t lds1 = lds;
    const size_t lds = (size_t)AS_SDPA_NBUF2 * 2 * GL_TILE;
    (void)lds1;
    static std::atomic<bool> attr_set{false};   // (idempotent attribute call: a race only repeats it)
    if (!attr_set && lds > 64 * 1024 - 1) {
      (void)hipFuncSetAttribute((const void*)sdpa_fwd_pipe_kernel<2, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)sdpa_fwd_pipe_kernel<2, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    if (impl == 2) AS_PIPE_LAUNCH(2, false, 0, grid2, s, 0, 1, nullptr);
    else AS_PIPE_LAUNCH(2, false, 1, grid2, s, 0, 1, nullptr);
    AS_CHECK_LAUNCH("sdpa_fwd_pipe<2>");
    return AS_OK;
  }
  if (ns > 0) {
    if (impl == 1) AS_PIPE_LAUNCH(1, true, 0, ns * BH, s2, qtiles, ns, ws);
    else if (impl == 3) AS_PIPE_LAUNCH(1, true, 1, ns * BH, s2, qtiles, ns, ws);
    else
      hipLaunchKernelGGL(sdpa_fwd_glds_kernel<true>, dim3(ns * BH), dim3(SD_NT), lds, s2, (const __bf16*)q, (const __bf16*)k,
                         (const __bf16*)vt, (__bf16*)o, lse, B, N, Npad, h, qtiles, ns, (float*)ws);
    AS_CHECK_LAUNCH("sdpa_fwd<split>");
    hipLaunchKernelGGL(sdpa_combine_kernel, dim3(SD_QB / 16, BH), dim3(256), 0, s2, (const float*)ws, (__bf16*)o, lse, B, N,
                       h, qtiles, ns);
    AS_CHECK_LAUNCH("sdpa_combine");
    if (s2 != s) (void)hipEventRecord(side.join, s2);
  }
  if (qtiles > 0) {
    if (impl == 1) AS_PIPE_LAUNCH(1, false, 0, qtiles * BH, s, 0, 1, nullptr);
    else if (impl == 3) AS_PIPE_LAUNCH(1, false, 1, qtiles * BH, s, 0, 1, nullptr);
    else
      hipLaunchKernelGGL(sdpa_fwd_glds_kernel<false>, dim3(qtiles * BH), dim3(SD_NT), lds, s, (const __bf16*)q,
                         (const __bf16*)k, (const __bf16*)vt, (__bf16*)o, lse, B, N, Npad, h, 0, 1, (float*)nullptr);
    AS_CHECK_LAUNCH("sdpa_fwd");
  }
#undef AS_PIPE_LAUNCH
#undef AS_PIPE_LAUNCH_MIX
  if (ns > 0 && s2 != s) (void)hipStreamWaitEvent(s, side.join, 0);     // join: later work on `s` sees the split rows
  return AS_OK;
}

template <typename T>
int launch_sdpa(const void* q, const void* k, const void* vt, void* o, float* lse, int B, int N, int h,
                hipStream_t s) {
  const int Npad = as_round_up(N, 64);
  const int grid = as_ceil_div(N, SD_QB) * B * h;
  const size_t lds = 2 * ((size_t)SD_KB * SdpaCfg<T>::K_PITCH + (size_t)HD * SdpaCfg<T>::V_PITCH);
  static std::atomic<bool> attr_set{false};   // (idempotent attribute call: a race only repeats it)
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)sdpa_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((sdpa_fwd_kernel<T>), dim3(grid), dim3(SD_NT), lds, s, (const T*)q, (const T*)k,
                     (const T*)vt, (T*)o, lse, B, N, Npad, h);
  AS_CHECK_LAUNCH("sdpa_fwd");
  return AS_OK;
}

}  // namespace

extern "C" size_t as_sdpa_fwd_workspace_bytes(int B, int N, int h, int dtype) {
  if (B <= 0 || N <= 0 || h <= 0 || dtype != AS_BF16) return 0;
  int ns = sdpa_split_slices(B, N, h);
  SdpaMix mix;
  if (sdpa_mix_plan(B, N, h, &mix) && N % SD_QB != 0) ns = ns > sdpa_tail_slices(B, N, h) ? ns : sdpa_tail_slices(B, N, h);
  size_t bytes = (size_t)B * h * ns * SD_QB * SD_REC * sizeof(float);
  if (sdpa_sk_ok(B, N, h) && sdpa_sk_ws_bytes(B, N, h) > bytes) bytes = sdpa_sk_ws_bytes(B, N, h);
  return bytes;
}

extern "C" int as_sdpa_fwd(const void* q, const void* k, const void* vt, void* o, float* lse, void* workspace,
                           size_t workspace_bytes, int B, int N, int h, int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && vt && o && lse, AS_E_BADARG, "as_sdpa_fwd: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && h > 0, AS_E_BADARG, "as_sdpa_fwd: bad sizes B=%d N=%d h=%d", B, N, h);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_sdpa_glds(q, k, vt, o, lse, workspace, workspace_bytes, B, N, h, s);
  if (dtype == AS_F32) return launch_sdpa<float>(q, k, vt, o, lse, B, N, h, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_sdpa_fwd: dtype %d", dtype);
}

extern "C" int as_attn_fwd(const void* x, const void* Wqkv, const float* bqkv, const void* Wproj,
                           const float* bproj, void* out, float* lse, void* q, void* k, void* vt, void* o,
                           void* workspace, size_t workspace_bytes, int B, int N, int D, int h, int dtype,
                           as_stream_t stream) {
  int rc = as_qkv_fwd(x, Wqkv, bqkv, q, k, vt, B, N, D, h, dtype, stream);
  if (rc != AS_OK) return rc;
  rc = as_sdpa_fwd(q, k, vt, o, lse, workspace, workspace_bytes, B, N, h, dtype, stream);
  if (rc != AS_OK) return rc;
  return as_linear_fwd(o, Wproj, bproj, out, B * N, D, D, dtype, 0, stream);
}
