// Backward of the fused scaled-dot-product attention (head dim 64) for gfx950: dQ, dK, dV from dO, recomputing the
// softmax tiles from q, k and the saved row log-sum-exp instead of reading a stored [h,N,N] matrix.
//
// Replaces the autograd backward of   attn = softmax(q k^T * d^-0.5); x = attn @ v
// (reference models/vision_transformer.py:79-83; under `use_checkpoint` the reference recomputes the forward of every
// block in backward, visual_transformer_det.py:232-236 -- the recompute here is per tile and stays on chip).
//
//   P  = exp(S * scale - lse)            S = Q K^T
//   dV = P^T dO
//   dP = dO V^T          delta_i = sum_d dO[i,d] O[i,d]
//   dS = P o (dP - delta) * scale
//   dQ = dS K            dK = dS^T Q
//
// Three kernels, no atomics, fixed summation order (bit-reproducible run to run):
//   bwd_prep      : delta, lse in base-2 units, row-major v; for the bf16 kernels of round 4 (sdpa_bwd_*_tr_kernel, the
//                   default) zero-padded row-major q / k / dO, for the fp32 parity path and AS_BWD_TR=0 a fragment-major
//                   copy of dO and the transposed copies (q^T, k^T, dO^T) those kernels need as k-contiguous A operands.
//   bwd_dq        : one workgroup = 128 queries (a lane owns ONE query), loops over 64-key tiles.
//   bwd_dkv       : one workgroup = 128 keys    (a lane owns ONE key),   loops over 64-query tiles.
// As in the forward every MFMA is "swapped" so the lane that owns a query (key) keeps its column through the whole
// chain, and the row operand of the first MFMA is fed in the order pi(i) = i with bits 2,3 swapped, which makes the
// accumulator registers 8*s .. 8*s+7 of a lane hold 8 CONSECUTIVE rows: P / dS go accumulator -> B operand of the
// next MFMA without leaving the lane, and that MFMA's A operand is a plain contiguous fragment of the transposed tile.
// Templated on the element type: bf16 (v_mfma_f32_32x32x16_bf16) and fp32 (exact v_mfma_f32_32x32x2_f32 chains; the
// parity path).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int BW_NT = 256, BW_HD = 64, BW_TILE = 64;
typedef __attribute__((ext_vector_type(4))) unsigned bw_u32x4;   // plain vector type: staging arrays stay in VGPRs

__device__ __forceinline__ int pi_row(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }
// accumulator register r of a lane in half `half` <-> row inside a 32-row block when the rows were fed through pi
__device__ __forceinline__ int pi_acc_row(int r, int half) { return 16 * (r >> 3) + 8 * half + (r & 7); }

__device__ __forceinline__ void st4(__bf16* p, float a, float b, float c, float d) {
  bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  *reinterpret_cast<bf16x4*>(p) = v;
}
// (cdna_hip_programming.md T21) A 64-column bf16 row held as acc[2][16] (lane: columns db * 32 + 8 g + 4 * half + r of ITS row,
// both half-waves on the same row): one v_permlane32_swap per dword and pair of column groups leaves lanes 0-31 with the 16
// contiguous bytes of group k and lanes 32-63 with those of group k + 1 -- eight 16-byte stores instead of sixteen 8-byte ones,
// same bytes, same addresses.  Every lane of the wave must call it; `valid` gates the stores only.
__device__ __forceinline__ void store_row64_wide(__bf16* row, const f32x16 (&acc)[2], float scale, int half, bool valid) {
  uint2 o2[8];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x4 v = {(__bf16)(acc[db][4 * g] * scale), (__bf16)(acc[db][4 * g + 1] * scale), (__bf16)(acc[db][4 * g + 2] * scale),
                        (__bf16)(acc[db][4 * g + 3] * scale)};
      o2[db * 4 + g] = __builtin_bit_cast(uint2, v);
    }
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    auto rx = __builtin_amdgcn_permlane32_swap(o2[k].x, o2[k + 1].x, false, false);
    auto ry = __builtin_amdgcn_permlane32_swap(o2[k].y, o2[k + 1].y, false, false);
    o2[k].x = rx[0]; o2[k + 1].x = rx[1];
    o2[k].y = ry[0]; o2[k + 1].y = ry[1];
  }
  if (valid) {
#pragma unroll
    for (int k = 0; k < 8; k += 2)
      *reinterpret_cast<uint4*>(row + 8 * k + 8 * half) = make_uint4(o2[k].x, o2[k].y, o2[k + 1].x, o2[k + 1].y);
  }
}
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// One 64-row x 64-element tile staged global -> registers -> LDS in 16-byte chunks by 256 threads.
template <typename T> struct TileStage {
  static constexpr int ROWB = BW_HD * (int)sizeof(T);       // bytes per tile row
  static constexpr int CPR = ROWB / 16;                     // chunks per row
  static constexpr int CH = BW_TILE * CPR / BW_NT;          // chunks per thread
};
template <typename T>
__device__ __forceinline__ void tile_load(bw_u32x4 (&r)[TileStage<T>::CH], const char* src, size_t src_row_stride, int tid) {
  constexpr int CPR = TileStage<T>::CPR;
#pragma unroll
  for (int i = 0; i < TileStage<T>::CH; ++i) {
    const int c = tid + i * BW_NT;
    r[i] = *reinterpret_cast<const bw_u32x4*>(src + (size_t)(c / CPR) * src_row_stride + (c % CPR) * 16);
  }
}
template <typename T>
__device__ __forceinline__ void tile_store(const bw_u32x4 (&r)[TileStage<T>::CH], char* dst, int pitch, int tid) {
  constexpr int CPR = TileStage<T>::CPR;
#pragma unroll
  for (int i = 0; i < TileStage<T>::CH; ++i) {
    const int c = tid + i * BW_NT;
    *reinterpret_cast<bw_u32x4*>(dst + (c / CPR) * pitch + (c % CPR) * 16) = r[i];
  }
}

template <typename T> __device__ __forceinline__ void lds_frag(Frag<T>& f, const char* p) {
  f.load16B(reinterpret_cast<const T*>(p));
}

// ---------------------------------------------------------------------------------------------------------
// prep: one workgroup per (image*head, 64-row tile, layout job)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BW_NT) void bwd_prep_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ vt, const T* __restrict__ o,
                                                         const T* __restrict__ d_o, T* __restrict__ dof,
                                                         T* __restrict__ dot, T* __restrict__ qt, T* __restrict__ kt,
                                                         T* __restrict__ vrow, float* __restrict__ delta,
                                                         const float* __restrict__ lse, float* __restrict__ lse2, int B, int N,
                                                         int Npad, int h) {
  __shared__ float tile[64][65];
  const int BH = B * h;
  const int bh = blockIdx.x % BH, t = blockIdx.x / BH;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x;
  const int D = h * BW_HD;
  const int row0 = t * 64;
  // blockIdx.y picks ONE of the four independent layout jobs of the tile (the dO job also writes delta and lse2): four
  // times the workgroups in flight instead of four LDS round trips in a row per workgroup
  const int job = blockIdx.y;
  if (job == 0) {
  // lse in base-2 units for the LDS-DMA kernels (an LDS-DMA cannot convert on the way): +inf on the padded rows, so P = 0
  if (tid < 64) lse2[(size_t)bh * Npad + row0 + tid] = row0 + tid < N ? lse[(size_t)bh * N + row0 + tid] * AS_LOG2E : INFINITY;

  // dO tile: rows -> dof (fragment-major), delta; transposed -> dot
  for (int e = tid; e < 64 * 64; e += BW_NT) {
    const int r = e >> 6, d = e & 63, n = row0 + r;
    float g = 0.0f, ov = 0.0f;
    if (n < N) {
      g = to_f32<T>(d_o[((size_t)b * N + n) * D + head * BW_HD + d]);
      ov = to_f32<T>(o[((size_t)b * N + n) * D + head * BW_HD + d]);
    }
    dof[qf_elem((size_t)bh, Npad, n, d)] = from_f32<T>(g);
    tile[r][d] = g;
    // row reduction of g * o inside the 64-lane wave that owns row r (e>>6 is wave-uniform: 64 consecutive e)
    const float s = wave_sum(g * ov);
    if (d == 0) delta[(size_t)bh * Npad + n] = s;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += BW_NT) {
    const int d = e >> 6, r = e & 63;
    dot[((size_t)bh * BW_HD + d) * Npad + row0 + r] = from_f32<T>(tile[r][d]);
  }
  } else if (job == 1) {
  // q (fragment-major) -> qt
  for (int e = tid; e < 64 * 64; e += BW_NT) {
    const int r = e >> 6, d = e & 63, n = row0 + r;
    tile[r][d] = n < N ? to_f32<T>(q[qf_elem((size_t)bh, Npad, n, d)]) : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += BW_NT) {
    const int d = e >> 6, r = e & 63;
    qt[((size_t)bh * BW_HD + d) * Npad + row0 + r] = from_f32<T>(tile[r][d]);
  }
  } else if (job == 2) {
  // k (row-major) -> kt
  for (int e = tid; e < 64 * 64; e += BW_NT) {
    const int r = e >> 6, d = e & 63, n = row0 + r;
    tile[r][d] = n < N ? to_f32<T>(k[((size_t)bh * Npad + n) * BW_HD + d]) : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += BW_NT) {
    const int d = e >> 6, r = e & 63;
    kt[((size_t)bh * BW_HD + d) * Npad + row0 + r] = from_f32<T>(tile[r][d]);
  }
  } else {
  // vt (transposed) -> v row-major, zero padded
  for (int e = tid; e < 64 * 64; e += BW_NT) {
    const int d = e >> 6, r = e & 63, n = row0 + r;
    tile[r][d] = n < N ? to_f32<T>(vt[((size_t)bh * BW_HD + d) * Npad + n]) : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += BW_NT) {
    const int r = e >> 6, d = e & 63;
    vrow[((size_t)bh * Npad + row0 + r) * BW_HD + d] = from_f32<T>(tile[r][d]);
  }
  }
}

// The same four jobs for bf16 with 16 bytes per lane on both sides of every transposition (the scalar template above
// moves 2 bytes per lane: 130 MB in 53 us).  A 64 x 64 tile is loaded as 512 vectors of 8 elements (row a, chunk cv),
// staged in LDS as packed pairs [a][b / 2] (pitch 33 words), and written as 512 vectors of 8 consecutive a for one b: a
// lane of the store phase reads its 8 values from one word column (8 row groups x 4 word columns per wave: 32 banks).
// Layout-only outputs (dof: the fragment-major copy of dO) are written straight from the load registers.
__device__ __forceinline__ void prep_stage(uint32_t (*tile)[33], int v, uint4 d) {
  const int row = v >> 3, cv = v & 7;
  tile[row][cv * 4 + 0] = d.x; tile[row][cv * 4 + 1] = d.y; tile[row][cv * 4 + 2] = d.z; tile[row][cv * 4 + 3] = d.w;
}
__device__ __forceinline__ uint4 prep_gather(const uint32_t (*tile)[33], int v) {   // vector (b = v >> 3, a = 8 * (v & 7) ..)
  const int bcol = v >> 3, ch = v & 7, w = bcol >> 1, sh = (bcol & 1) * 16;
  uint32_t e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = (tile[ch * 8 + j][w] >> sh) & 0xffffu;
  return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

// ROWM (round 4, the kernels without transposed copies): dO, q, k are written ROW-MAJOR [bh][Npad][64] with zero rows
// beyond N (into the dof / qt / kt slots of the workspace); nothing is transposed but v.
template <bool ROWM>
__global__ __launch_bounds__(BW_NT) void bwd_prep_vec_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                             const __bf16* __restrict__ vt, const __bf16* __restrict__ o,
                                                             const __bf16* __restrict__ d_o, __bf16* __restrict__ dof,
                                                             __bf16* __restrict__ dot, __bf16* __restrict__ qt,
                                                             __bf16* __restrict__ kt, __bf16* __restrict__ vrow,
                                                             float* __restrict__ delta, const float* __restrict__ lse,
                                                             float* __restrict__ lse2, int B, int N, int Npad, int h) {
  __shared__ uint32_t tile[64][33];
  const int BH = B * h;
  const int bh = blockIdx.x % BH, t = blockIdx.x / BH;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x;
  const int D = h * BW_HD;
  const int row0 = t * 64;
  const int job = blockIdx.y;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  if (job == 0) {
    // (ROWM: both row statistics are stored NEGATED -- they enter the score MFMAs as C operands)
    if (tid < 64) {
      const float l2 = row0 + tid < N ? lse[(size_t)bh * N + row0 + tid] * AS_LOG2E : INFINITY;
      lse2[(size_t)bh * Npad + row0 + tid] = ROWM ? -l2 : l2;
    }
    // dO rows: -> dof (fragment-major: the 8 elements of a (row, d-chunk) are one 16-byte run there too), delta, LDS
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + BW_NT * i, r = v >> 3, cv = v & 7, n = row0 + r;
      uint4 g = zero4, ov = zero4;
      if (n < N) {
        g = *reinterpret_cast<const uint4*>(d_o + ((size_t)b * N + n) * D + head * BW_HD + cv * 8);
        ov = *reinterpret_cast<const uint4*>(o + ((size_t)b * N + n) * D + head * BW_HD + cv * 8);
      }
      if (ROWM) *reinterpret_cast<uint4*>(dof + ((size_t)bh * Npad + n) * BW_HD + cv * 8) = g;
      else {
        *reinterpret_cast<uint4*>(dof + qf_elem((size_t)bh, Npad, n, cv * 8)) = g;
        prep_stage(tile, v, g);
      }
      const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        s += __uint_as_float(gw[j] << 16) * __uint_as_float(ow[j] << 16) +
             __uint_as_float(gw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
      // the 8 lanes of a row are consecutive: three xor steps inside the group of 8 (fixed order)
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      if (cv == 0) delta[(size_t)bh * Npad + n] = ROWM ? -s : s;
    }
    if (ROWM) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + BW_NT * i;
      *reinterpret_cast<uint4*>(dot + ((size_t)bh * BW_HD + (v >> 3)) * Npad + row0 + (v & 7) * 8) = prep_gather(tile, v);
    }
  } else if (job == 1 || job == 2) {
    // q (fragment-major) -> qt, k (row-major) -> kt: rows n, transposed to [d][n]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + BW_NT * i, r = v >> 3, cv = v & 7, n = row0 + r;
      uint4 d = zero4;
      if (n < N)
        d = job == 1 ? *reinterpret_cast<const uint4*>(q + qf_elem((size_t)bh, Npad, n, cv * 8))
                     : *reinterpret_cast<const uint4*>(k + ((size_t)bh * Npad + n) * BW_HD + cv * 8);
      if (ROWM) *reinterpret_cast<uint4*>((job == 1 ? qt : kt) + ((size_t)bh * Npad + n) * BW_HD + cv * 8) = d;
      else prep_stage(tile, v, d);
    }
    if (ROWM) return;
    __syncthreads();
    __bf16* dst = job == 1 ? qt : kt;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + BW_NT * i;
      *reinterpret_cast<uint4*>(dst + ((size_t)bh * BW_HD + (v >> 3)) * Npad + row0 + (v & 7) * 8) = prep_gather(tile, v);
    }
  } else {
    // vt [d][n] -> v rows [n][d], zero beyond N: the tile's rows are d here, 8 consecutive n per vector
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + BW_NT * i, d = v >> 3, cv = v & 7, n = row0 + cv * 8;
      uint4 x = *reinterpret_cast<const uint4*>(vt + ((size_t)bh * BW_HD + d) * Npad + n);
      if (n + 8 > N) {                                   // ragged end: keep the elements below N
        uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (n + 2 * j >= N) w[j] = 0u;
          else if (n + 2 * j + 1 >= N) w[j] &= 0xffffu;
        }
        x = make_uint4(w[0], w[1], w[2], w[3]);
      }
      prep_stage(tile, v, x);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + BW_NT * i;                     // vector: n = row0 + (v >> 3), d = 8 * (v & 7) ..
      *reinterpret_cast<uint4*>(vrow + ((size_t)bh * Npad + row0 + (v >> 3)) * BW_HD + (v & 7) * 8) = prep_gather(tile, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// dQ: workgroup = 128 queries of one (image, head); key tiles of 64
//   S^T  = K . Q^T        A = K rows (pi order)     B = Q^T  (registers)
//   dP^T = V . dO^T       A = V rows (pi order)     B = dO^T (registers)
//   dQ^T += K^T . dS^T    A = K^T rows d            B = dS^T (from the accumulators)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BW_NT, sizeof(T) == 2 ? 2 : 1) void sdpa_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ dof,
                                                            const T* __restrict__ k, const T* __restrict__ vrow,
                                                            const T* __restrict__ kt, const float* __restrict__ lse,
                                                            const float* __restrict__ delta, T* __restrict__ dqkv,
                                                            int B, int N, int Npad, int h) {
  using TS = TileStage<T>;
  constexpr int ES = (int)sizeof(T);
  constexpr int PITCH = TS::ROWB + 16;
  constexpr int TILE_B = BW_TILE * PITCH;
  constexpr int BUF_B = 3 * TILE_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int BH = B * h;
  const int bh = blockIdx.x % BH, qtile = blockIdx.x / BH;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int query = qtile * 128 + wave * 32 + li;
  const int qc = min(query, N - 1);

  Frag<T> fq[4], fdo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fq[ks].load16B(q + qf_frag((size_t)bh, Npad, qc, ks, half));
    fdo[ks].load16B(dof + qf_frag((size_t)bh, Npad, qc, ks, half));
  }
  const float lse2 = lse[(size_t)bh * N + qc] * 1.44269504088896340736f;
  const float dl = delta[(size_t)bh * Npad + qc];

  const char* ksrc = reinterpret_cast<const char*>(k + (size_t)bh * Npad * BW_HD);
  const char* vsrc = reinterpret_cast<const char*>(vrow + (size_t)bh * Npad * BW_HD);
  const char* ktsrc = reinterpret_cast<const char*>(kt + (size_t)bh * BW_HD * Npad);
  bw_u32x4 sk[TS::CH], sv[TS::CH], skt[TS::CH];
#define GLOAD(t)                                                                         \
  do {                                                                                   \
    tile_load<T>(sk, ksrc + (size_t)(t) * BW_TILE * TS::ROWB, TS::ROWB, tid);            \
    tile_load<T>(sv, vsrc + (size_t)(t) * BW_TILE * TS::ROWB, TS::ROWB, tid);            \
    tile_load<T>(skt, ktsrc + (size_t)(t) * BW_TILE * ES, (size_t)Npad * ES, tid);       \
  } while (0)
#define LSTORE(buf)                                                                      \
  do {                                                                                   \
    char* base_ = smem + (buf) * BUF_B;                                                  \
    tile_store<T>(sk, base_, PITCH, tid);                                                \
    tile_store<T>(sv, base_ + TILE_B, PITCH, tid);                                       \
    tile_store<T>(skt, base_ + 2 * TILE_B, PITCH, tid);                                  \
  } while (0)

  f32x16 dqacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dqacc[0][r] = 0.0f; dqacc[1][r] = 0.0f; }

  const int nkt = Npad / BW_TILE;
  const int prow = pi_row(li);
  GLOAD(0);
  LSTORE(0);
  __syncthreads();
  for (int t = 0; t < nkt; ++t) {
    if (t + 1 < nkt) GLOAD(t + 1);
    const char* Ks = smem + (t & 1) * BUF_B;
    const char* Vs = Ks + TILE_B;
    const char* Kts = Ks + 2 * TILE_B;
    Frag<T> fds[2][2];
    const bool ragged = (t + 1) * BW_TILE > N;
    auto scores = [&](auto ragged_c) {                 // (masking compiled into the last key tile's copy only)
    constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.0f; pacc[r] = 0.0f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<T> fa;
        lds_frag(fa, Ks + (kb * 32 + prow) * PITCH + (ks * 16 + half * 8) * ES);
        sacc = mma32(fa, fq[ks], sacc);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<T> fa;
        lds_frag(fa, Vs + (kb * 32 + prow) * PITCH + (ks * 16 + half * 8) * ES);
        pacc = mma32(fa, fdo[ks], pacc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(sacc[r] - lse2);     // q pre-scaled (common.h): base-2 logits
        float ds = p * (pacc[r] - dl);                  // the softmax scale 1/8 is applied once, to the accumulators
        if (RAGGED && t * BW_TILE + kb * 32 + pi_acc_row(r, half) >= N) ds = 0.0f;   // padded key rows hold garbage
        fds[kb][r >> 3].set(r & 7, ds);
      }
    }
    };
    if (ragged) scores(std::true_type{});
    else scores(std::false_type{});
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          Frag<T> fa;
          lds_frag(fa, Kts + (db * 32 + li) * PITCH + (kb * 32 + s2 * 16 + half * 8) * ES);
          dqacc[db] = mma32(fa, fds[kb][s2], dqacc[db]);
        }
    if (t + 1 < nkt) LSTORE((t + 1) & 1);
    __syncthreads();
  }
#undef GLOAD
#undef LSTORE

  if (query < N) {
    T* row = dqkv + ((size_t)b * N + query) * (size_t)(3 * h * BW_HD) + head * BW_HD;       // q slot of [3,h,64]
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st4(row + db * 32 + 8 * g + 4 * half, dqacc[db][4 * g] * 0.125f, dqacc[db][4 * g + 1] * 0.125f,
            dqacc[db][4 * g + 2] * 0.125f, dqacc[db][4 * g + 3] * 0.125f);
  }
}

// ---------------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 128 keys of one (image, head); query tiles of 64
//   S    = Q . K^T        A = Q rows (pi order; the fragment-major q tile is already in operand order)
//   dP   = dO . V^T       A = dO rows (pi order, fragment-major copy)        B = K, V fragments (registers)
//   dV^T += dO^T . P      A = dO^T rows d      B = P  (from the accumulators)
//   dK^T += Q^T . dS      A = Q^T rows d       B = dS (from the accumulators)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BW_NT, sizeof(T) == 2 ? 2 : 1) void sdpa_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ dof,
                                                             const T* __restrict__ qt, const T* __restrict__ dot,
                                                             const T* __restrict__ k, const T* __restrict__ vrow,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dqkv,
                                                             int B, int N, int Npad, int h) {
  using TS = TileStage<T>;
  constexpr int ES = (int)sizeof(T);
  constexpr int PITCH = TS::ROWB + 16;
  constexpr int FRAG_B = BW_TILE * TS::ROWB;          // fragment-major tiles are copied linearly (no padding)
  constexpr int TR_B = BW_TILE * PITCH;
  constexpr int BUF_B = 2 * FRAG_B + 2 * TR_B + 2 * BW_TILE * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int BH = B * h;
  const int bh = blockIdx.x % BH, ktile = blockIdx.x / BH;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int key = ktile * 128 + wave * 32 + li;
  const int kc = min(key, Npad - 1);

  Frag<T> fk[4], fv[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fk[ks].load16B(k + ((size_t)bh * Npad + kc) * BW_HD + ks * 16 + half * 8);
    fv[ks].load16B(vrow + ((size_t)bh * Npad + kc) * BW_HD + ks * 16 + half * 8);
  }

  const char* qsrc = reinterpret_cast<const char*>(q + (size_t)bh * Npad * BW_HD);
  const char* dosrc = reinterpret_cast<const char*>(dof + (size_t)bh * Npad * BW_HD);
  const char* qtsrc = reinterpret_cast<const char*>(qt + (size_t)bh * BW_HD * Npad);
  const char* dotsrc = reinterpret_cast<const char*>(dot + (size_t)bh * BW_HD * Npad);
  bw_u32x4 sq[TS::CH], sdo[TS::CH], sqt[TS::CH], sdot[TS::CH];
  float stat = 0.0f;                                 // threads 0..63: lse*log2e, 64..127: delta
#define GLOAD(t)                                                                                                   \
  do {                                                                                                             \
    tile_load<T>(sq, qsrc + (size_t)(t) * FRAG_B, TS::ROWB, tid);                                                  \
    tile_load<T>(sdo, dosrc + (size_t)(t) * FRAG_B, TS::ROWB, tid);                                                \
    tile_load<T>(sqt, qtsrc + (size_t)(t) * BW_TILE * ES, (size_t)Npad * ES, tid);                                 \
    tile_load<T>(sdot, dotsrc + (size_t)(t) * BW_TILE * ES, (size_t)Npad * ES, tid);                               \
    if (tid < 64) {                                                                                                \
      const int n_ = (t) * BW_TILE + tid; /* padded query: P = exp2(-inf) = 0 */                                   \
      stat = n_ < N ? lse[(size_t)bh * N + n_] * 1.44269504088896340736f : INFINITY;                               \
    } else if (tid < 128) {                                                                                        \
      const int n_ = (t) * BW_TILE + tid - 64;                                                                     \
      stat = n_ < N ? delta[(size_t)bh * Npad + n_] : 0.0f;                                                        \
    }                                                                                                              \
  } while (0)
#define LSTORE(buf)                                                                                                \
  do {                                                                                                             \
    char* base_ = smem + (buf) * BUF_B;                                                                            \
    tile_store<T>(sq, base_, TS::ROWB, tid);                                                                       \
    tile_store<T>(sdo, base_ + FRAG_B, TS::ROWB, tid);                                                             \
    tile_store<T>(sqt, base_ + 2 * FRAG_B, PITCH, tid);                                                            \
    tile_store<T>(sdot, base_ + 2 * FRAG_B + TR_B, PITCH, tid);                                                    \
    if (tid < 128) reinterpret_cast<float*>(base_ + 2 * FRAG_B + 2 * TR_B)[tid] = stat;                            \
  } while (0)

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dkacc[0][r] = 0.0f; dkacc[1][r] = 0.0f; dvacc[0][r] = 0.0f; dvacc[1][r] = 0.0f; }

  const int nqt = Npad / BW_TILE;
  const int prow = pi_row(li);
  GLOAD(0);
  LSTORE(0);
  __syncthreads();
  for (int t = 0; t < nqt; ++t) {
    if (t + 1 < nqt) GLOAD(t + 1);
    const char* Qs = smem + (t & 1) * BUF_B;
    const char* dOs = Qs + FRAG_B;
    const char* Qts = Qs + 2 * FRAG_B;
    const char* dOts = Qts + TR_B;
    const float* st = reinterpret_cast<const float*>(dOts + TR_B);
    const bool ragged = (t + 1) * BW_TILE > N;

    Frag<T> fp[2][2], fds[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 sacc, pacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.0f; pacc[r] = 0.0f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<T> fa;
        lds_frag(fa, Qs + (size_t)(((qb * 4 + ks) * 64) + prow + 32 * half) * 8 * ES);
        sacc = mma32(fa, fk[ks], sacc);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<T> fa;
        lds_frag(fa, dOs + (size_t)(((qb * 4 + ks) * 64) + prow + 32 * half) * 8 * ES);
        pacc = mma32(fa, fv[ks], pacc);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int q0 = qb * 32 + 16 * s2 + 8 * half;            // 8 consecutive queries of this lane's registers
        const float4 l0 = *reinterpret_cast<const float4*>(st + q0), l1 = *reinterpret_cast<const float4*>(st + q0 + 4);
        const float4 d0 = *reinterpret_cast<const float4*>(st + 64 + q0),
                     d1 = *reinterpret_cast<const float4*>(st + 64 + q0 + 4);
        const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
          const int r = s2 * 8 + t8;
          float p = __builtin_amdgcn_exp2f(sacc[r] - lv[t8]);        // q pre-scaled (common.h): base-2 logits
          float ds = p * (pacc[r] - dv[t8]);            // scale applied once to the dK accumulators
          if (ragged && t * BW_TILE + q0 + t8 >= N) { p = 0.0f; ds = 0.0f; }   // padded query rows hold garbage
          fp[qb][s2].set(t8, p);
          fds[qb][s2].set(t8, ds);
        }
      }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          Frag<T> fa;
          lds_frag(fa, dOts + (db * 32 + li) * PITCH + (qb * 32 + s2 * 16 + half * 8) * ES);
          dvacc[db] = mma32(fa, fp[qb][s2], dvacc[db]);
          lds_frag(fa, Qts + (db * 32 + li) * PITCH + (qb * 32 + s2 * 16 + half * 8) * ES);
          dkacc[db] = mma32(fa, fds[qb][s2], dkacc[db]);
        }
    if (t + 1 < nqt) LSTORE((t + 1) & 1);
    __syncthreads();
  }
#undef GLOAD
#undef LSTORE

  if (key < N) {
    const int D = h * BW_HD;
    T* row = dqkv + ((size_t)b * N + key) * (size_t)(3 * D) + head * BW_HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * half;
        // dK = dS^T q / 8 with the STORED q' = q log2(e) / 8: dS^T q' ln 2
        st4(row + D + d, dkacc[db][4 * g] * AS_LN2, dkacc[db][4 * g + 1] * AS_LN2, dkacc[db][4 * g + 2] * AS_LN2,
            dkacc[db][4 * g + 3] * AS_LN2);
        st4(row + 2 * D + d, dvacc[db][4 * g], dvacc[db][4 * g + 1], dvacc[db][4 * g + 2], dvacc[db][4 * g + 3]);
      }
  }
}


// ---------------------------------------------------------------------------------------------------------
// dK, dV (bf16) on an LDS-DMA ring: the arithmetic of sdpa_bwd_dkv_kernel, with the query tile's four operand images
// (Q and dO fragment-major: 8 one-KiB lane-linear pieces each; Q^T and dO^T: 64 rows of 128 B, 8 pieces of 8 rows, the
// 16-byte chunks swizzled on the SOURCE side as in gemm.hip) and its 128 row statistics brought by `global_load_lds`
// straight into the ring -- no staging registers (32 fewer VGPRs), no per-thread address arithmetic in the loop, the loads
// of tile t+1 in flight under the MFMAs of tile t.  The DMA is issued from inline asm (scalar base + lane offset), so
// hipcc schedules the fragment reads as ordinary LDS loads; one raw barrier per tile.
// ---------------------------------------------------------------------------------------------------------
#ifndef AS_BWD_TR_DEFAULT
#define AS_BWD_TR_DEFAULT 1                  // transposing LDS reads instead of transposed operand copies
#endif
#ifndef AS_BWD_TR_NST_DKV
#define AS_BWD_TR_NST_DKV 2                  // ring depth (2, 3, 4 measure the same within 2 %: nothing waits for the DMA)
#endif
#ifndef AS_BWD_TR_NST_DQ
#define AS_BWD_TR_NST_DQ 2
#endif
#ifndef AS_BWD_NST
#define AS_BWD_NST 2                         // ring stages (2: 66.6 KB, two workgroups per CU; 3: 99.8 KB, one)
#endif
constexpr int DK_QF = 0, DK_DOF = 8192, DK_QT = 16384, DK_DOT = 24576, DK_ST = 32768, DK_STAGE = 33280;

__device__ __forceinline__ void bw_dma16(unsigned voff, const char* sbase, unsigned dst) {
  unsigned keep;   // global_load_lds_dwordx4, saddr form: 64 lanes x 16 B from sbase + voff[lane] -> LDS [dst + 16 * lane]
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
__device__ __forceinline__ void bw_dma4(unsigned voff, const char* sbase, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
__device__ __forceinline__ int bw_swz(int r) { return ((r >> 1) & 3) | (((r >> 4) & 1) << 2); }   // gemm.hip g_swz<4>

// NK blocks of 32 keys per wave (every fragment read of the query tile feeds NK MFMAs), NW waves per workgroup (each staged
// query tile serves 32 * NK * NW keys) -- see sdpa_bwd_dq_dma_kernel for the LDS-pipe arithmetic behind both.  A key's sums
// run over the same tiles in the same order for every shape: bitwise the same dK / dV.
template <int NK, int NW>
__global__ __launch_bounds__(64 * NW, 2) void sdpa_bwd_dkv_dma_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ dof,
                                                                    const __bf16* __restrict__ qt, const __bf16* __restrict__ dot,
                                                                    const __bf16* __restrict__ k, const __bf16* __restrict__ vrow,
                                                                    const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                    __bf16* __restrict__ dqkv, int B, int N, int Npad, int h) {
  using T = __bf16;
  constexpr int KROWS = 32 * NK * NW, PPW = 8 / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int BH = B * h;
  const int bh = blockIdx.x % BH, ktile = blockIdx.x / BH;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  int key[NK];
  Frag<T> fk[NK][4], fv[NK][4];
#pragma unroll
  for (int nk = 0; nk < NK; ++nk) {
    key[nk] = ktile * KROWS + (wave * NK + nk) * 32 + li;
    const int kc = min(key[nk], Npad - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fk[nk][ks].load16B(k + ((size_t)bh * Npad + kc) * BW_HD + ks * 16 + half * 8);
      fv[nk][ks].load16B(vrow + ((size_t)bh * Npad + kc) * BW_HD + ks * 16 + half * 8);
    }
  }
  // the ordinary loads are complete before the first LDS-DMA is issued (hipcc does not count the asm DMAs in vmcnt)
#pragma unroll
  for (int nk = 0; nk < NK; ++nk)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(fk[nk][0].v), "+v"(fk[nk][1].v), "+v"(fk[nk][2].v), "+v"(fk[nk][3].v), "+v"(fv[nk][0].v),
                 "+v"(fv[nk][1].v), "+v"(fv[nk][2].v), "+v"(fv[nk][3].v));

  const unsigned smem_u = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  const char* qf_b = reinterpret_cast<const char*>(q + (size_t)bh * Npad * BW_HD);
  const char* dof_b = reinterpret_cast<const char*>(dof + (size_t)bh * Npad * BW_HD);
  const char* qt_b = reinterpret_cast<const char*>(qt + (size_t)bh * BW_HD * Npad);
  const char* dot_b = reinterpret_cast<const char*>(dot + (size_t)bh * BW_HD * Npad);
  const char* l2_b = reinterpret_cast<const char*>(lse2 + (size_t)bh * Npad);
  const char* dl_b = reinterpret_cast<const char*>(delta + (size_t)bh * Npad);
  const unsigned voff = lane * 16;
  unsigned voff_t[PPW];                      // my 8-row pieces of the transposed tiles: row stride Npad, swizzled chunk
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int r = (wave * PPW + j) * 8 + (lane >> 3);
    voff_t[j] = (unsigned)r * (unsigned)Npad * 2u + (unsigned)(((lane & 7) ^ bw_swz(r)) << 4);
  }
  auto stage = [&](int t, int slot) {
    const unsigned base = smem_u + slot * DK_STAGE;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int p = wave * PPW + j;
      bw_dma16(voff, qf_b + (size_t)t * 8192 + p * 1024, base + DK_QF + p * 1024);
      bw_dma16(voff, dof_b + (size_t)t * 8192 + p * 1024, base + DK_DOF + p * 1024);
      bw_dma16(voff_t[j], qt_b + (size_t)t * 128, base + DK_QT + p * 1024);
      bw_dma16(voff_t[j], dot_b + (size_t)t * 128, base + DK_DOT + p * 1024);
    }
    if (wave == 0) bw_dma4(lane * 4, l2_b + (size_t)t * 256, base + DK_ST);
    if (wave == 1) bw_dma4(lane * 4, dl_b + (size_t)t * 256, base + DK_ST + 256);
  };
  auto ring_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  f32x16 dkacc[NK][2], dvacc[NK][2];
#pragma unroll
  for (int nk = 0; nk < NK; ++nk)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkacc[nk][0][r] = 0.0f; dkacc[nk][1][r] = 0.0f; dvacc[nk][0][r] = 0.0f; dvacc[nk][1][r] = 0.0f; }

  const int nqt = Npad / BW_TILE;
  const int prow = pi_row(li);
  const int sw = bw_swz(li);
  int off_t[4];                              // [qb * 2 + s2]: byte offset of this lane's chunk in row li of a transposed tile
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) off_t[c2] = li * 128 + (((c2 * 2 + half) ^ sw) << 4);

#pragma unroll
  for (int p_ = 0; p_ < AS_BWD_NST - 1; ++p_)
    if (p_ < nqt) stage(p_, p_);
  for (int t = 0; t < nqt; ++t) {
    // my pieces of tile t have landed: everything but the (AS_BWD_NST - 2) younger tiles (4 * PPW DMAs per wave and tile, one
    // more on the two waves that carry the row statistics)
    if (AS_BWD_NST == 2 || t + 1 >= nqt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPW + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPW) : "memory");
    ring_barrier();                          // tile t is complete for everyone; everyone is done reading tile t-1
    if (t + AS_BWD_NST - 1 < nqt) stage(t + AS_BWD_NST - 1, (t + AS_BWD_NST - 1) % AS_BWD_NST);
    const char* Qs = smem + (t % AS_BWD_NST) * DK_STAGE;
    const char* dOs = Qs + DK_DOF;
    const char* Qts = Qs + DK_QT;
    const char* dOts = Qs + DK_DOT;
    const float* st = reinterpret_cast<const float*>(Qs + DK_ST);
    const bool ragged = (t + 1) * BW_TILE > N;

    // (the masking of the padded query rows is compiled into the LAST tile's copy of the block only: left as a run-time
    // condition hipcc if-converts it into 128 v_cndmask + 32 v_cmp per tile, 44 % of the loop's VALU)
    auto tile = [&](auto ragged_c) {
    constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      Frag<T> fp[NK][2], fds[NK][2];
      {
        f32x16 sacc[NK], pacc[NK];
#pragma unroll
        for (int nk = 0; nk < NK; ++nk)
#pragma unroll
          for (int r = 0; r < 16; ++r) { sacc[nk][r] = 0.0f; pacc[nk][r] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          Frag<T> fa;
          lds_frag(fa, Qs + (size_t)(((qb * 4 + ks) * 64) + prow + 32 * half) * 16);
#pragma unroll
          for (int nk = 0; nk < NK; ++nk) sacc[nk] = mma32(fa, fk[nk][ks], sacc[nk]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          Frag<T> fa;
          lds_frag(fa, dOs + (size_t)(((qb * 4 + ks) * 64) + prow + 32 * half) * 16);
#pragma unroll
          for (int nk = 0; nk < NK; ++nk) pacc[nk] = mma32(fa, fv[nk][ks], pacc[nk]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int q0 = qb * 32 + 16 * s2 + 8 * half;            // 8 consecutive queries of this lane's registers
          const float4 l0 = *reinterpret_cast<const float4*>(st + q0), l1 = *reinterpret_cast<const float4*>(st + q0 + 4);
          const float4 d0 = *reinterpret_cast<const float4*>(st + 64 + q0),
                       d1 = *reinterpret_cast<const float4*>(st + 64 + q0 + 4);
          const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
          const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
          for (int nk = 0; nk < NK; ++nk)
#pragma unroll
            for (int t8 = 0; t8 < 8; ++t8) {
              const int r = s2 * 8 + t8;
              float p = __builtin_amdgcn_exp2f(sacc[nk][r] - lv[t8]);        // q pre-scaled (common.h): base-2 logits
              float ds = p * (pacc[nk][r] - dv[t8]);        // scale applied once to the dK accumulators
              if (RAGGED && t * BW_TILE + q0 + t8 >= N) { p = 0.0f; ds = 0.0f; }   // padded query rows hold garbage
              fp[nk][s2].set(t8, p);
              fds[nk][s2].set(t8, ds);
            }
        }
      }
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          Frag<T> fa;
          lds_frag(fa, dOts + db * 4096 + off_t[qb * 2 + s2]);
#pragma unroll
          for (int nk = 0; nk < NK; ++nk) dvacc[nk][db] = mma32(fa, fp[nk][s2], dvacc[nk][db]);
          lds_frag(fa, Qts + db * 4096 + off_t[qb * 2 + s2]);
#pragma unroll
          for (int nk = 0; nk < NK; ++nk) dkacc[nk][db] = mma32(fa, fds[nk][s2], dkacc[nk][db]);
        }
    }
    };
    if (ragged) tile(std::true_type{});
    else tile(std::false_type{});
  }

#pragma unroll
  for (int nk = 0; nk < NK; ++nk)
    if (key[nk] < N) {
      const int D = h * BW_HD;
      T* row = dqkv + ((size_t)b * N + key[nk]) * (size_t)(3 * D) + head * BW_HD;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = db * 32 + 8 * g + 4 * half;
          // dK = dS^T q / 8 with the STORED q' = q log2(e) / 8: dS^T q' ln 2
          st4(row + D + d, dkacc[nk][db][4 * g] * AS_LN2, dkacc[nk][db][4 * g + 1] * AS_LN2, dkacc[nk][db][4 * g + 2] * AS_LN2,
              dkacc[nk][db][4 * g + 3] * AS_LN2);
          st4(row + 2 * D + d, dvacc[nk][db][4 * g], dvacc[nk][db][4 * g + 1], dvacc[nk][db][4 * g + 2], dvacc[nk][db][4 * g + 3]);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// dQ (bf16) on an LDS-DMA ring: the arithmetic of sdpa_bwd_dq_kernel; per 64-key tile the K rows, the V rows (both 64 rows
// of 128 B, contiguous in HBM) and the K^T tile (64 rows d of 128 B, row stride Npad) arrive as 8 + 8 + 8 one-KiB pieces
// of 8 rows with the 16-byte chunks swizzled on the source side; two stages of 24 KiB.
// ---------------------------------------------------------------------------------------------------------
constexpr int DQ_K = 0, DQ_V = 8192, DQ_KT = 16384, DQ_STAGE = 24576;

// NQ blocks of 32 queries per wave (a lane owns NQ queries: every K / V / K^T fragment read from LDS feeds NQ MFMAs), NW
// waves per workgroup (each staged tile serves 32 * NQ * NW queries).  Why both: with one query per lane and four waves the
// loop issues one ds_read_b128 per MFMA (4 LDS cycles against the MFMA's 32 on a quarter of the CU: 50 % of the LDS pipe) and
// lands 24 KiB of DMA per 96 MFMAs (≈ 330 LDS cycles per 768 MFMA cycles: another 43 %) -- the LDS pipe, not the matrix
// pipe, was the busiest unit.  <2, 4>: 46 %; <2, 8>: 36 %.  The arithmetic of a query does not change (same tiles, same
// order): every shape gives bitwise the same dQ.
#ifndef AS_BWD_DQ_OCC
#define AS_BWD_DQ_OCC 2                    // waves per SIMD asked of hipcc (<1, 4>: 3 caps at 170 registers (2 spills) and measures 2-4 % slower)
#endif
template <int NQ, int NW>
__global__ __launch_bounds__(64 * NW, AS_BWD_DQ_OCC) void sdpa_bwd_dq_dma_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ dof,
                                                                   const __bf16* __restrict__ k, const __bf16* __restrict__ vrow,
                                                                   const __bf16* __restrict__ kt, const float* __restrict__ lse,
                                                                   const float* __restrict__ delta, __bf16* __restrict__ dqkv,
                                                                   int B, int N, int Npad, int h) {
  using T = __bf16;
  constexpr int QROWS = 32 * NQ * NW, PPW = 8 / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int BH = B * h;
  const int bh = blockIdx.x % BH, qtile = blockIdx.x / BH;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  int query[NQ];
  Frag<T> fq[NQ][4], fdo[NQ][4];
  float lse2[NQ], dl[NQ];
#pragma unroll
  for (int nq = 0; nq < NQ; ++nq) {
    query[nq] = qtile * QROWS + (wave * NQ + nq) * 32 + li;
    const int qc = min(query[nq], N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fq[nq][ks].load16B(q + qf_frag((size_t)bh, Npad, qc, ks, half));
      fdo[nq][ks].load16B(dof + qf_frag((size_t)bh, Npad, qc, ks, half));
    }
    lse2[nq] = lse[(size_t)bh * N + qc] * AS_LOG2E;
    dl[nq] = delta[(size_t)bh * Npad + qc];
  }
  // the ordinary loads are complete before the first LDS-DMA is issued (hipcc does not count the asm DMAs in vmcnt)
#pragma unroll
  for (int nq = 0; nq < NQ; ++nq)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(fq[nq][0].v), "+v"(fq[nq][1].v), "+v"(fq[nq][2].v), "+v"(fq[nq][3].v), "+v"(fdo[nq][0].v),
                 "+v"(fdo[nq][1].v), "+v"(fdo[nq][2].v), "+v"(fdo[nq][3].v), "+v"(lse2[nq]), "+v"(dl[nq]));

  const unsigned smem_u = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  const char* k_b = reinterpret_cast<const char*>(k + (size_t)bh * Npad * BW_HD);
  const char* v_b = reinterpret_cast<const char*>(vrow + (size_t)bh * Npad * BW_HD);
  const char* kt_b = reinterpret_cast<const char*>(kt + (size_t)bh * BW_HD * Npad);
  unsigned voff_r[PPW], voff_t[PPW];         // my 8-row pieces: of the row-major tiles (row stride 128 B) / of K^T (stride Npad)
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int r = (wave * PPW + j) * 8 + (lane >> 3);
    const unsigned ch = (unsigned)(((lane & 7) ^ bw_swz(r)) << 4);
    voff_r[j] = (unsigned)r * 128u + ch;
    voff_t[j] = (unsigned)r * (unsigned)Npad * 2u + ch;
  }
  auto stage = [&](int t, int slot) {
    const unsigned base = smem_u + slot * DQ_STAGE;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int p = wave * PPW + j;
      bw_dma16(voff_r[j], k_b + (size_t)t * 8192, base + DQ_K + p * 1024);
      bw_dma16(voff_r[j], v_b + (size_t)t * 8192, base + DQ_V + p * 1024);
      bw_dma16(voff_t[j], kt_b + (size_t)t * 128, base + DQ_KT + p * 1024);
    }
  };
  auto ring_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  f32x16 dqacc[NQ][2];
#pragma unroll
  for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dqacc[nq][0][r] = 0.0f; dqacc[nq][1][r] = 0.0f; }

  const int nkt = Npad / BW_TILE;
  const int prow = pi_row(li);
  int off_k[4], off_t[4];                    // [ks] in row prow of a row-major tile / [kb * 2 + s2] in row li of the K^T tile
  {
    const int swp = bw_swz(prow), swl = bw_swz(li);
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      off_k[c2] = prow * 128 + (((c2 * 2 + half) ^ swp) << 4);
      off_t[c2] = li * 128 + (((c2 * 2 + half) ^ swl) << 4);
    }
  }
  stage(0, 0);
  for (int t = 0; t < nkt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my pieces of tile t (the only DMA in flight)
    ring_barrier();                                           // tile t complete; everyone is done reading tile t-1
    if (t + 1 < nkt) stage(t + 1, (t + 1) & 1);
    const char* Ks = smem + (t & 1) * DQ_STAGE;
    const char* Vs = Ks + DQ_V;
    const char* Kts = Ks + DQ_KT;
    const bool ragged = (t + 1) * BW_TILE > N;
    auto tile = [&](auto ragged_c) {
      constexpr bool RAGGED = decltype(ragged_c)::value;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        Frag<T> fds[NQ][2];
        {
          f32x16 sacc[NQ], pacc[NQ];
#pragma unroll
          for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[nq][r] = 0.0f; pacc[nq][r] = 0.0f; }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            Frag<T> fa;
            lds_frag(fa, Ks + kb * 4096 + off_k[ks]);
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) sacc[nq] = mma32(fa, fq[nq][ks], sacc[nq]);
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            Frag<T> fa;
            lds_frag(fa, Vs + kb * 4096 + off_k[ks]);
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) pacc[nq] = mma32(fa, fdo[nq][ks], pacc[nq]);
          }
          if (NQ > 1) __builtin_amdgcn_sched_barrier(0);      // (no K^T fragment hoisted above the conversion: registers)
#pragma unroll
          for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float p = __builtin_amdgcn_exp2f(sacc[nq][r] - lse2[nq]);
              float ds = p * (pacc[nq][r] - dl[nq]);
              if (RAGGED && t * BW_TILE + kb * 32 + pi_acc_row(r, half) >= N) ds = 0.0f;   // padded key rows hold garbage
              fds[nq][r >> 3].set(r & 7, ds);
            }
        }
        if (NQ > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            Frag<T> fa;
            lds_frag(fa, Kts + db * 4096 + off_t[kb * 2 + s2]);
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) dqacc[nq][db] = mma32(fa, fds[nq][s2], dqacc[nq][db]);
          }
        if (NQ > 1) __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (ragged) tile(std::true_type{});
    else tile(std::false_type{});
  }

#pragma unroll
  for (int nq = 0; nq < NQ; ++nq)
    if (query[nq] < N) {
      T* row = dqkv + ((size_t)b * N + query[nq]) * (size_t)(3 * h * BW_HD) + head * BW_HD;       // q slot of [3,h,64]
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          st4(row + db * 32 + 8 * g + 4 * half, dqacc[nq][db][4 * g] * 0.125f, dqacc[nq][db][4 * g + 1] * 0.125f,
              dqacc[nq][db][4 * g + 2] * 0.125f, dqacc[nq][db][4 * g + 3] * 0.125f);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 4: the same two kernels WITHOUT transposed operand copies.  The products dV^T += dO^T P^T, dK^T += Q^T dS^T (and
// dQ^T += K^T dS^T) need the staged tile with the contraction index (query / key) along the fragment -- a COLUMN of the
// row-major tile.  gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns, lane c gets
// column c's four rows; two reads = one bf16x8 operand) takes it from the SAME image the row fragments of the score
// products are read from, so a stage holds two 8-KiB tiles instead of four (dK/dV) or three (dQ): half (two thirds of)
// the L2 -> LDS bytes, no strided 128-byte rows at stride 2 Npad, a 4-deep ring in the LDS of the old 2-deep one -- the
// PMC profile of the old kernels shows waves waiting 58 % (dK/dV) of their time, the MFMA pipe busy 32 %: each iteration
// cost one exposed DMA latency.  The prep kernel no longer writes q^T, k^T, dO^T.
//   image of a 64-row x 128-byte tile: row r at r * 128, its 16-byte chunk c at position c ^ t3_swz(r) (swizzled on the
//   DMA's source address).  t3_swz makes both access patterns bank-conflict-free: the ds_read_b128 row fragments (16-lane
//   service groups over rows {0-3,12-15,20-27} / {4-11,16-19,28-31}: the eight rows of one parity get eight distinct
//   positions) and the transposing reads (a 32-lane group covers 4 rows x 4 chunks: rows r, r + 2 land in different halves
//   of the 128-byte line).
// A row's sums run over the same tiles in the same order with the same operand values: bitwise the results of the kernels
// above.
// ---------------------------------------------------------------------------------------------------------
typedef short bw_i16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int t3_swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
// transposed fragment (lds_frag_tr3 below): lane (li, half) gets rows [8 half .. 8 half + 7] of column li of a 16-row x
// 32-column block from two reads (rows +0..3, rows +4..7)
template <int CNT> __device__ __forceinline__ void bw_wait_newer(int newer) {       // my DMAs of all but `newer` tiles have landed
  if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CNT) : "memory");
  else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
typedef __attribute__((address_space(3))) const char* bw_lds_ptr;       // 32-bit LDS pointers: lane pointer + constant folds into the DS immediate
struct T3Lane {                              // per-lane pointers into slot 0's first tile image
  bw_lds_ptr pa[4];                          // [ks]: row pi(li), chunk 2 ks + half (add 4096 for the second 32-row block)
  bw_lds_ptr pt[2][2];                       // [db][q]: transposing read of rows 4 q .. 4 q + 3 of a 16-row group, columns 32 db .. (add 4096 qb + 2048 s2)
  bw_lds_ptr pst;                            // my 8-row run of the row statistics (dK/dV kernel)
  unsigned voff[2];                          // DMA source offsets of my two 8-row pieces (row-major source, 128-byte rows)
};
__device__ __forceinline__ T3Lane t3_lane(const char* smem, int lane, int wave) {
  T3Lane L;
  const bw_lds_ptr base = (bw_lds_ptr)smem;
  const int li = lane & 31, half = lane >> 5, prow = pi_row(li);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) L.pa[ks] = base + (prow * 128 + (((2 * ks + half) ^ t3_swz(prow)) << 4));
  const int g = lane >> 4, tt = lane & 15;
  const int rl = 8 * (g >> 1) + (tt >> 2), cl = 2 * (g & 1) + ((tt & 3) >> 1);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = rl + 4 * q, c = 4 * db + cl;
      L.pt[db][q] = base + (r * 128 + ((c ^ t3_swz(r)) << 4) + (tt & 1) * 8);
    }
  L.pst = base + (16384 + 32 * half);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + (lane >> 3);
    L.voff[j] = (unsigned)(r * 128 + (((lane & 7) ^ t3_swz(r)) << 4));
  }
  return L;
}
__device__ __forceinline__ void lds_frag3(Frag<__bf16>& f, bw_lds_ptr p) {
  f.v = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(p);
}
__device__ __forceinline__ void lds_frag_tr3(Frag<__bf16>& f, bw_lds_ptr p0, bw_lds_ptr p1) {
  const bw_i16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bw_i16x4*)p0);
  const bw_i16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bw_i16x4*)p1);
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  const i16x8 w = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  f.v = __builtin_bit_cast(bf16x8, w);
}

constexpr int T3_A = 0, T3_B = 8192, T3_ST = 16384;
constexpr int T3_DKV_STAGE = 16896, T3_DQ_STAGE = 16384;
#ifndef AS_BWD_ABLATE
#define AS_BWD_ABLATE 0                      // (timing experiments only: 1 no DMA in the loop, 2 no barrier, 3 packs only, 4 no VALU in E)
#endif
constexpr int T3_ABL = AS_BWD_ABLATE;

// Issue order inside a 32-row block (both kernels).  Left to itself hipcc emits `ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma`
// pairs -- one exposed LDS latency per MFMA, 32 (24) per tile -- and spills the addresses of the transposing reads.  The
// blocks below are therefore written as phases fenced with sched_barrier(0): every LDS read of a phase is issued in one
// burst a phase ahead of its first use, and the matrix products of a phase run back to back on independent accumulators:
//   [row fragments + statistics of the block]  ->  C: 8 score MFMAs (S, dP alternating)  ->  [transposed fragments]  ->
//   E: exp2 / multiply / pack  ->  [row fragments + statistics of the NEXT block]  ->  F: the 8 (4) gradient MFMAs.
// The row statistics enter as the C operand of the first score MFMA (-lse2 for S, -delta for dP: the prep kernel stores
// them negated), so the element-wise part is exp2, one multiply and the packs.
__device__ __forceinline__ unsigned t3_pack(float a, float b) {      // two fp32 -> packed bf16 (RNE), one instruction
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void t3_frag_words(Frag<__bf16>& f, unsigned a, unsigned b, unsigned c, unsigned d) {
  const bw_u32x4 w = {a, b, c, d};
  f.v = __builtin_bit_cast(bf16x8, w);
}
template <typename F, int... Is> __device__ __forceinline__ void t3_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void t3_static_for(F&& f) {
  t3_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
#ifndef AS_BWD_PRIO
#define AS_BWD_PRIO 0                        // 1: s_setprio 1 around the MFMA phases (experiments)
#endif
__device__ __forceinline__ void t3_prio(int p) {
  if (AS_BWD_PRIO) { if (p) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
}
struct T3Rows { Frag<__bf16> a[4], b[4]; };
template <int OFF> __device__ __forceinline__ void t3_rows(T3Rows& R, const T3Lane& L) {     // OFF: slot + 4096 * block; second tile at + 8192
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    lds_frag3(R.a[ks], L.pa[ks] + OFF);
    lds_frag3(R.b[ks], L.pa[ks] + (OFF + T3_B));
  }
}
template <int OFF> __device__ __forceinline__ f32x16 t3_stat16(const T3Lane& L) {       // 8 + 8 consecutive values of the two 16-row groups
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) f4* f4p;
  const f4 a0 = *(f4p)(L.pst + OFF), a1 = *(f4p)(L.pst + (OFF + 16)), a2 = *(f4p)(L.pst + (OFF + 64)), a3 = *(f4p)(L.pst + (OFF + 80));
  const f32x16 v = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3], a3[0], a3[1], a3[2], a3[3]};
  return v;
}

template <int NST>
__global__ __launch_bounds__(BW_NT, 2) void sdpa_bwd_dkv_tr_kernel(const __bf16* __restrict__ qr, const __bf16* __restrict__ dor,
                                                                   const __bf16* __restrict__ k, const __bf16* __restrict__ vrow,
                                                                   const float* __restrict__ nlse2, const float* __restrict__ ndelta,
                                                                   __bf16* __restrict__ dqkv, int B, int N, int Npad, int h) {
  using T = __bf16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int BH = B * h;
  const int bh = blockIdx.x % BH, ktile = blockIdx.x / BH;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int key = ktile * 128 + wave * 32 + li;
  const int kc = min(key, Npad - 1);

  Frag<T> fk[4], fv[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fk[ks].load16B(k + ((size_t)bh * Npad + kc) * BW_HD + ks * 16 + half * 8);
    fv[ks].load16B(vrow + ((size_t)bh * Npad + kc) * BW_HD + ks * 16 + half * 8);
  }
  // (these ordinary loads are waited for BEHIND the first tiles' LDS-DMA below: one round trip instead of two)

  const unsigned smem_u = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  const char* q_b = reinterpret_cast<const char*>(qr + (size_t)bh * Npad * BW_HD);
  const char* do_b = reinterpret_cast<const char*>(dor + (size_t)bh * Npad * BW_HD);
  const char* l2_b = reinterpret_cast<const char*>(nlse2 + (size_t)bh * Npad);
  const char* dl_b = reinterpret_cast<const char*>(ndelta + (size_t)bh * Npad);
  const T3Lane L = t3_lane(smem, lane, wave);
  auto stage = [&](int t, int slot) {
    const unsigned base = smem_u + slot * T3_DKV_STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = wave * 2 + j;
      bw_dma16(L.voff[j], q_b + (size_t)t * 8192, base + T3_A + p * 1024);
      bw_dma16(L.voff[j], do_b + (size_t)t * 8192, base + T3_B + p * 1024);
    }
    if (wave == 0) bw_dma4(lane * 4, l2_b + (size_t)t * 256, base + T3_ST);
    if (wave == 1) bw_dma4(lane * 4, dl_b + (size_t)t * 256, base + T3_ST + 256);
  };

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dkacc[0][r] = 0.0f; dkacc[1][r] = 0.0f; dvacc[0][r] = 0.0f; dvacc[1][r] = 0.0f; }

  const int nqt = Npad / BW_TILE;
#pragma unroll
  for (int p_ = 0; p_ < NST - 1; ++p_)
    if (p_ < nqt) stage(p_, p_);
  // the K / V fragments (ordinary loads issued at the top) and the tiles just staged land together; hipcc does not count the asm
  // DMAs, so its wait for the fragments must be complete HERE, before the loop's counted waits (the loads are older than every
  // DMA and retire first, but a compiler-placed vmcnt(0) at their first use inside the loop would drain the ring every tile)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(fk[0].v), "+v"(fk[1].v), "+v"(fk[2].v), "+v"(fk[3].v), "+v"(fv[0].v), "+v"(fv[1].v),
               "+v"(fv[2].v), "+v"(fv[3].v));
  // the ring slot is a compile-time constant inside the body (NST tiles per trip): every LDS address is a loop-invariant
  // lane offset + an immediate
  for (int t0 = 0; t0 < nqt; t0 += NST) {
    t3_static_for<NST>([&](auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    const int t = t0 + SLOT;
    if (t >= nqt) return;
    const int newer = min(NST - 2, nqt - 1 - t);
    if (wave < 2) bw_wait_newer<5>(newer);
    else bw_wait_newer<4>(newer);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (T3_ABL != 2) __builtin_amdgcn_s_barrier();   // tile t is complete for everyone; everyone is done reading tile t-1
    asm volatile("" ::: "memory");
    if (T3_ABL != 1 && t + NST - 1 < nqt) stage(t + NST - 1, (SLOT + NST - 1) % NST);
    constexpr int QO = SLOT * T3_DKV_STAGE;   // Q tile of the slot; dO tile at + T3_B, statistics at T3_ST
    // (no masking of the padded query rows: their q and dO rows are zero and their -lse2 is -inf, so P = dS = 0)

    T3Rows R;
    f32x16 sacc, pacc;
    sacc = t3_stat16<QO>(L);
    pacc = t3_stat16<QO + 256>(L);
    t3_rows<QO>(R, L);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // C: S^T = K Q^T - lse2, dP^T = V dO^T - delta
      t3_prio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        sacc = mma32(R.a[ks], fk[ks], sacc);
        pacc = mma32(R.b[ks], fv[ks], pacc);
      }
      __builtin_amdgcn_sched_barrier(0);
      Frag<T> tdo[2][2], tq[2][2];             // [db][s2]: dO^T / Q^T fragments of this block
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int o = QO + qb * 4096 + s2 * 2048;
          lds_frag_tr3(tdo[db][s2], L.pt[db][0] + (o + T3_B), L.pt[db][1] + (o + T3_B));
          lds_frag_tr3(tq[db][s2], L.pt[db][0] + o, L.pt[db][1] + o);
        }
      t3_prio(0);
      __builtin_amdgcn_sched_barrier(0);
      // E: P = exp2(S'), dS = P dP' (q pre-scaled, common.h: base-2 logits; the 1/8 is applied once to the dK sums)
      Frag<T> fp[2], fds[2];
      {
        unsigned wp[8], wd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (T3_ABL == 3) { wp[j] = t3_pack(sacc[2 * j], sacc[2 * j + 1]); wd[j] = t3_pack(pacc[2 * j], pacc[2 * j + 1]); continue; }
          if (T3_ABL == 4) { wp[j] = __float_as_uint(sacc[2 * j]); wd[j] = __float_as_uint(pacc[2 * j + 1]); continue; }
          const float p0 = __builtin_amdgcn_exp2f(sacc[2 * j]), p1 = __builtin_amdgcn_exp2f(sacc[2 * j + 1]);
          wp[j] = t3_pack(p0, p1);
          wd[j] = t3_pack(p0 * pacc[2 * j], p1 * pacc[2 * j + 1]);
        }
        t3_frag_words(fp[0], wp[0], wp[1], wp[2], wp[3]);
        t3_frag_words(fp[1], wp[4], wp[5], wp[6], wp[7]);
        t3_frag_words(fds[0], wd[0], wd[1], wd[2], wd[3]);
        t3_frag_words(fds[1], wd[4], wd[5], wd[6], wd[7]);
      }
      asm volatile("" : "+v"(fp[0].v), "+v"(fp[1].v), "+v"(fds[0].v), "+v"(fds[1].v));
      __builtin_amdgcn_sched_barrier(0);
      if (qb == 0) {
        sacc = t3_stat16<QO + 128>(L);
        pacc = t3_stat16<QO + 256 + 128>(L);
        t3_rows<QO + 4096>(R, L);
        __builtin_amdgcn_sched_barrier(0);
      }
      // F: dV^T += dO^T P^T, dK^T += Q^T dS^T
      t3_prio(1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dvacc[db] = mma32(tdo[db][s2], fp[s2], dvacc[db]);
          dkacc[db] = mma32(tq[db][s2], fds[s2], dkacc[db]);
        }
      t3_prio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    });
  }

  {
    const int D = h * BW_HD;
    const int keyc = min(key, N - 1);
    T* row = dqkv + ((size_t)b * N + keyc) * (size_t)(3 * D) + head * BW_HD;
    // dK = dS^T q / 8 with the STORED q' = q log2(e) / 8: dS^T q' ln 2
    store_row64_wide(row + D, dkacc, AS_LN2, half, key < N);
    store_row64_wide(row + 2 * D, dvacc, 1.0f, half, key < N);
  }
}

template <int NST>
__global__ __launch_bounds__(BW_NT, 2) void sdpa_bwd_dq_tr_kernel(const __bf16* __restrict__ qr, const __bf16* __restrict__ dor,
                                                                  const __bf16* __restrict__ kr, const __bf16* __restrict__ vrow,
                                                                  const float* __restrict__ lse, const float* __restrict__ ndelta,
                                                                  __bf16* __restrict__ dqkv, int B, int N, int Npad, int h) {
  using T = __bf16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int BH = B * h;
  const int bh = blockIdx.x % BH, qtile = blockIdx.x / BH;
  const int b = bh / h, head = bh % h;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int query = qtile * 128 + wave * 32 + li;
  const int qc = min(query, N - 1);

  Frag<T> fq[4], fdo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fq[ks].load16B(qr + ((size_t)bh * Npad + qc) * BW_HD + ks * 16 + half * 8);
    fdo[ks].load16B(dor + ((size_t)bh * Npad + qc) * BW_HD + ks * 16 + half * 8);
  }
  float nl2 = -lse[(size_t)bh * N + qc] * AS_LOG2E;
  float ndl = ndelta[(size_t)bh * Npad + qc];
  // (waited for behind the first tiles' LDS-DMA below: one round trip instead of two)

  const unsigned smem_u = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  const char* k_b = reinterpret_cast<const char*>(kr + (size_t)bh * Npad * BW_HD);
  const char* v_b = reinterpret_cast<const char*>(vrow + (size_t)bh * Npad * BW_HD);
  const T3Lane L = t3_lane(smem, lane, wave);
  auto stage = [&](int t, int slot) {
    const unsigned base = smem_u + slot * T3_DQ_STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = wave * 2 + j;
      bw_dma16(L.voff[j], k_b + (size_t)t * 8192, base + T3_A + p * 1024);
      bw_dma16(L.voff[j], v_b + (size_t)t * 8192, base + T3_B + p * 1024);
    }
  };

  f32x16 dqacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dqacc[0][r] = 0.0f; dqacc[1][r] = 0.0f; }

  const int nkt = Npad / BW_TILE;
#pragma unroll
  for (int p_ = 0; p_ < NST - 1; ++p_)
    if (p_ < nkt) stage(p_, p_);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(fq[0].v), "+v"(fq[1].v), "+v"(fq[2].v), "+v"(fq[3].v), "+v"(fdo[0].v), "+v"(fdo[1].v),
               "+v"(fdo[2].v), "+v"(fdo[3].v), "+v"(nl2), "+v"(ndl));
  // my query's statistics as C operands of the first score MFMAs (every accumulator register belongs to my query)
  f32x16 c_l, c_d;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c_l[r] = nl2; c_d[r] = ndl; }
  for (int t0 = 0; t0 < nkt; t0 += NST) {
    t3_static_for<NST>([&](auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    const int t = t0 + SLOT;
    if (t >= nkt) return;
    bw_wait_newer<4>(min(NST - 2, nkt - 1 - t));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (T3_ABL != 2) __builtin_amdgcn_s_barrier();   // tile t complete; everyone is done reading tile t-1
    asm volatile("" ::: "memory");
    if (T3_ABL != 1 && t + NST - 1 < nkt) stage(t + NST - 1, (SLOT + NST - 1) % NST);
    constexpr int KO = SLOT * T3_DQ_STAGE;    // K tile of the slot; V tile at + T3_B
    const bool ragged = (t + 1) * BW_TILE > N;
    auto tile = [&](auto ragged_c) {
      constexpr bool RAGGED = decltype(ragged_c)::value;
      T3Rows R;
      t3_rows<KO>(R, L);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        // C: S^T = K Q^T - lse2, dP^T = V dO^T - delta
        t3_prio(1);
        f32x16 sacc = mma32(R.a[0], fq[0], c_l), pacc = mma32(R.b[0], fdo[0], c_d);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
          sacc = mma32(R.a[ks], fq[ks], sacc);
          pacc = mma32(R.b[ks], fdo[ks], pacc);
        }
        __builtin_amdgcn_sched_barrier(0);
        Frag<T> tk[2][2];                      // [db][s2]: K^T fragments of this block
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            const int o = KO + kb * 4096 + s2 * 2048;
            lds_frag_tr3(tk[db][s2], L.pt[db][0] + o, L.pt[db][1] + o);
          }
        if (kb == 0) t3_rows<KO + 4096>(R, L);
        t3_prio(0);
        __builtin_amdgcn_sched_barrier(0);
        // E: dS = exp2(S') dP'
        Frag<T> fds[2];
        {
          unsigned wd[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (T3_ABL == 3) { wd[j] = t3_pack(sacc[2 * j] + pacc[2 * j], sacc[2 * j + 1] + pacc[2 * j + 1]); continue; }
            if (T3_ABL == 4) { wd[j] = __float_as_uint(sacc[2 * j]) ^ __float_as_uint(pacc[2 * j + 1]); continue; }
            float d0 = __builtin_amdgcn_exp2f(sacc[2 * j]) * pacc[2 * j], d1 = __builtin_amdgcn_exp2f(sacc[2 * j + 1]) * pacc[2 * j + 1];
            if (RAGGED) {                      // padded key rows
              if (t * BW_TILE + kb * 32 + pi_acc_row(2 * j, half) >= N) d0 = 0.0f;
              if (t * BW_TILE + kb * 32 + pi_acc_row(2 * j + 1, half) >= N) d1 = 0.0f;
            }
            wd[j] = t3_pack(d0, d1);
          }
          t3_frag_words(fds[0], wd[0], wd[1], wd[2], wd[3]);
          t3_frag_words(fds[1], wd[4], wd[5], wd[6], wd[7]);
        }
        asm volatile("" : "+v"(fds[0].v), "+v"(fds[1].v));
        __builtin_amdgcn_sched_barrier(0);
        // F: dQ^T += K^T dS^T
        t3_prio(1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int db = 0; db < 2; ++db) dqacc[db] = mma32(tk[db][s2], fds[s2], dqacc[db]);
        t3_prio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (ragged) tile(std::true_type{});
    else tile(std::false_type{});
    });
  }

  {
    T* row = dqkv + ((size_t)b * N + min(query, N - 1)) * (size_t)(3 * h * BW_HD) + head * BW_HD;       // q slot of [3,h,64]
    store_row64_wide(row, dqacc, 0.125f, half, query < N);
  }
}

template <typename T> size_t bwd_ws_bytes(int B, int N, int h) {
  const size_t Npad = as_round_up(N, 64);
  return (size_t)B * h * Npad * (5 * BW_HD * sizeof(T) + 2 * sizeof(float));      // ... + delta + lse2
}

template <typename T>
int launch_bwd(const void* q, const void* k, const void* vt, const void* o, const void* d_o, const float* lse,
               void* dqkv, void* ws, int B, int N, int h, hipStream_t s) {
  using TS = TileStage<T>;
  const int Npad = as_round_up(N, 64);
  const size_t per = (size_t)B * h * Npad * BW_HD;
  T* dof = (T*)ws;
  T* dot = dof + per;
  T* qt = dot + per;
  T* kt = qt + per;
  T* vrow = kt + per;
  float* delta = (float*)(vrow + per);
  float* lse2 = delta + (size_t)B * h * Npad;
  const int BH = B * h;
  static const bool prep_scalar = getenv("AS_BWD_PREP_SCALAR") != nullptr;      // (experiments: the 2-byte template)
  // AS_BWD_TR=0: round 3's kernels with transposed operand copies (qt / kt / dot) instead of the transposing LDS reads
  static const bool use_tr = !(getenv("AS_BWD_TR") && atoi(getenv("AS_BWD_TR")) == 0) && AS_BWD_TR_DEFAULT;
  if (sizeof(T) == 2 && !prep_scalar && use_tr && !getenv("AS_BWD_OLD"))
    hipLaunchKernelGGL(bwd_prep_vec_kernel<true>, dim3(BH * (Npad / 64), 4), dim3(BW_NT), 0, s, (const __bf16*)q, (const __bf16*)k,
                       (const __bf16*)vt, (const __bf16*)o, (const __bf16*)d_o, (__bf16*)dof, (__bf16*)dot, (__bf16*)qt,
                       (__bf16*)kt, (__bf16*)vrow, delta, lse, lse2, B, N, Npad, h);
  else if (sizeof(T) == 2 && !prep_scalar)
    hipLaunchKernelGGL(bwd_prep_vec_kernel<false>, dim3(BH * (Npad / 64), 4), dim3(BW_NT), 0, s, (const __bf16*)q, (const __bf16*)k,
                       (const __bf16*)vt, (const __bf16*)o, (const __bf16*)d_o, (__bf16*)dof, (__bf16*)dot, (__bf16*)qt,
                       (__bf16*)kt, (__bf16*)vrow, delta, lse, lse2, B, N, Npad, h);
  else
    hipLaunchKernelGGL((bwd_prep_kernel<T>), dim3(BH * (Npad / 64), 4), dim3(BW_NT), 0, s, (const T*)q, (const T*)k,
                       (const T*)vt, (const T*)o, (const T*)d_o, dof, dot, qt, kt, vrow, delta, lse, lse2, B, N, Npad, h);
  AS_CHECK_LAUNCH("sdpa_bwd_prep");
  constexpr int PITCH = TS::ROWB + 16;
  const size_t lds_dq = 2 * (size_t)(3 * BW_TILE * PITCH);
  const size_t lds_dkv = 2 * (size_t)(2 * BW_TILE * TS::ROWB + 2 * BW_TILE * PITCH + 2 * BW_TILE * 4);
  static std::atomic<bool> attr_set{false};   // (idempotent attribute call: a race only repeats it)
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)sdpa_bwd_dq_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq);
    (void)hipFuncSetAttribute((const void*)sdpa_bwd_dkv_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dkv);
    attr_set = true;
  }
  const int tiles = as_ceil_div(N, 128);
  static const bool old_dkv = getenv("AS_BWD_OLD") != nullptr;       // (experiments: the register-staged kernel)
  // dQ depends on the prep kernel only and writes its own third of dqkv: the bf16 kernels run it CONCURRENTLY with dK/dV
  // on a helper stream (fork after prep, join on the caller's stream).  (common.h AsSide.)
  static thread_local AsSide side;
  hipStream_t sq = s;
  if (sizeof(T) == 2 && !old_dkv) sq = as_side_fork(side, s);
  if constexpr (sizeof(T) == 2) {
    if (!old_dkv) {
      const size_t lds_dkv_dma = (size_t)AS_BWD_NST * DK_STAGE, lds_dq_dma = 2 * (size_t)DQ_STAGE;
      auto go_dkv = [&](auto nk_c, auto nw_c) {
        constexpr int NK = decltype(nk_c)::value, NW = decltype(nw_c)::value;
        static std::atomic<bool> attr{false};
        if (!attr) {
          (void)hipFuncSetAttribute((const void*)sdpa_bwd_dkv_dma_kernel<NK, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_dkv_dma);
          attr = true;
        }
        hipLaunchKernelGGL((sdpa_bwd_dkv_dma_kernel<NK, NW>), dim3(BH * as_ceil_div(N, 32 * NK * NW)), dim3(64 * NW), lds_dkv_dma, s,
                           (const __bf16*)q, (const __bf16*)dof, (const __bf16*)qt, (const __bf16*)dot, (const __bf16*)k,
                           (const __bf16*)vrow, (const float*)lse2, (const float*)delta, (__bf16*)dqkv, B, N, Npad, h);
      };
      auto go_dq = [&](auto nq_c, auto nw_c) {
        constexpr int NQ = decltype(nq_c)::value, NW = decltype(nw_c)::value;
        static std::atomic<bool> attr{false};
        if (!attr) {
          (void)hipFuncSetAttribute((const void*)sdpa_bwd_dq_dma_kernel<NQ, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_dq_dma);
          attr = true;
        }
        hipLaunchKernelGGL((sdpa_bwd_dq_dma_kernel<NQ, NW>), dim3(BH * as_ceil_div(N, 32 * NQ * NW)), dim3(64 * NW), lds_dq_dma, sq,
                           (const __bf16*)q, (const __bf16*)dof, (const __bf16*)k, (const __bf16*)vrow, (const __bf16*)kt, lse,
                           (const float*)delta, (__bf16*)dqkv, B, N, Npad, h);
      };
      if (use_tr && !prep_scalar) {
        constexpr int NSTK = AS_BWD_TR_NST_DKV, NSTQ = AS_BWD_TR_NST_DQ;
        const size_t lds_k = (size_t)NSTK * T3_DKV_STAGE, lds_q = (size_t)NSTQ * T3_DQ_STAGE;
        static std::atomic<bool> attr{false};
        if (!attr) {
          (void)hipFuncSetAttribute((const void*)sdpa_bwd_dkv_tr_kernel<NSTK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
          (void)hipFuncSetAttribute((const void*)sdpa_bwd_dq_tr_kernel<NSTQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
          attr = true;
        }
        // (workspace slots under ROWM: dof = dO rows, qt = q rows, kt = k rows)
        static const int only = getenv("AS_BWD_ONLY") ? atoi(getenv("AS_BWD_ONLY")) : 0;      // (timing experiments: 1 dK/dV, 2 dQ, 3 the prep kernel alone)
        if (only != 2 && only != 3)
        hipLaunchKernelGGL((sdpa_bwd_dkv_tr_kernel<NSTK>), dim3(BH * tiles), dim3(BW_NT), lds_k, s, (const __bf16*)qt, (const __bf16*)dof,
                           (const __bf16*)k, (const __bf16*)vrow, (const float*)lse2, (const float*)delta, (__bf16*)dqkv, B, N, Npad, h);
        AS_CHECK_LAUNCH("sdpa_bwd_dkv_tr");
        if (only != 1 && only != 3)
        hipLaunchKernelGGL((sdpa_bwd_dq_tr_kernel<NSTQ>), dim3(BH * tiles), dim3(BW_NT), lds_q, sq, (const __bf16*)qt, (const __bf16*)dof,
                           (const __bf16*)kt, (const __bf16*)vrow, lse, (const float*)delta, (__bf16*)dqkv, B, N, Npad, h);
        AS_CHECK_LAUNCH("sdpa_bwd_dq_tr");
        as_side_join(side, sq, s);
        return AS_OK;
      }
      // (round 4 tried other workgroup shapes of these two kernels -- 8 waves sharing a staged tile, 64 rows per wave:
      //  bitwise the same results, 4-6 % and 60 % (spills) slower: the loop was never short of LDS bandwidth, it exposed one
      //  LDS latency per MFMA; profiles/r04_attn_bwd.md)
      go_dkv(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
      AS_CHECK_LAUNCH("sdpa_bwd_dkv");
      go_dq(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
      as_side_join(side, sq, s);                          // the caller's stream continues after dQ as well
    }
  }
  if (sizeof(T) != 2 || old_dkv)
    hipLaunchKernelGGL((sdpa_bwd_dkv_kernel<T>), dim3(BH * tiles), dim3(BW_NT), lds_dkv, s, (const T*)q, dof, qt, dot,
                       (const T*)k, vrow, lse, delta, (T*)dqkv, B, N, Npad, h);
  AS_CHECK_LAUNCH("sdpa_bwd_dkv");
  if (sizeof(T) != 2 || old_dkv)
    hipLaunchKernelGGL((sdpa_bwd_dq_kernel<T>), dim3(BH * tiles), dim3(BW_NT), lds_dq, s, (const T*)q, dof,
                       (const T*)k, vrow, kt, lse, delta, (T*)dqkv, B, N, Npad, h);
  AS_CHECK_LAUNCH("sdpa_bwd_dq");
  return AS_OK;
}

}  // namespace

extern "C" size_t as_sdpa_bwd_workspace_bytes(int B, int N, int h, int dtype) {
  if (B <= 0 || N <= 0 || h <= 0) return 0;
  return dtype == AS_F32 ? bwd_ws_bytes<float>(B, N, h) : bwd_ws_bytes<__bf16>(B, N, h);
}

extern "C" int as_sdpa_bwd(const void* q, const void* k, const void* vt, const void* o, const void* d_o,
                           const float* lse, void* dqkv, void* workspace, size_t workspace_bytes, int B, int N, int h,
                           int dtype, as_stream_t stream) {
  AS_REQUIRE(q && k && vt && o && d_o && lse && dqkv && workspace, AS_E_BADARG, "as_sdpa_bwd: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && h > 0, AS_E_BADARG, "as_sdpa_bwd: bad sizes B=%d N=%d h=%d", B, N, h);
  AS_REQUIRE(dtype == AS_F32 || dtype == AS_BF16, AS_E_UNSUPPORTED, "as_sdpa_bwd: dtype %d", dtype);
  AS_REQUIRE(workspace_bytes >= as_sdpa_bwd_workspace_bytes(B, N, h, dtype), AS_E_BADARG,
             "as_sdpa_bwd: workspace too small (%zu < %zu)", workspace_bytes, as_sdpa_bwd_workspace_bytes(B, N, h, dtype));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_bwd<__bf16>(q, k, vt, o, d_o, lse, dqkv, workspace, B, N, h, s);
  return launch_bwd<float>(q, k, vt, o, d_o, lse, dqkv, workspace, B, N, h, s);
}
