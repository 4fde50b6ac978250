// Weight gradient of an nn.Linear WITHOUT transposed copies of the activations (round 4):
//
//     dW[n][k] = sum_m dY[m][n] * X[m][k]            dY [M, Nout], X [M, K] row-major bf16, M = tokens (8 k .. 51 k rows)
//
// Both operands of this product are contracted over their ROW index, so a k-contiguous MFMA fragment (8 consecutive m for
// one feature) is a COLUMN of a row-major tile.  Rounds 2-3 materialised dY^T and X^T first (transpose_pad_vec: 2.2 ms of
// the 39 ms training step).  Here the tiles go into LDS as they are -- 32 token rows x 128 features, lane-linear LDS-DMA,
// 16-byte chunks swizzled on the source address -- and the fragments come out through gfx950's transposing LDS read,
// ds_read_b64_tr_b16: every 16-lane group reads a 4-row x 16-column block (each lane supplies the address of its 8-byte
// piece) and lane c receives column c's four rows; two such reads are one bf16x8 operand (probe:
// tools/experiments/tr_probe.hip).  Split over the token range like as_linear_splitk_fwd: fp32 partials
// [S][Nout][K] summed in range order by splitk_reduce (no atomics).  Workgroup = 4 waves (2 x 2), tile 128 (n) x 128 (k),
// a wave owns 64 x 64; D[k][n] orientation so that a lane owns a row n of dW and its registers run along k (16-byte
// stores).  Replaces the dy^T / x^T branch of as_linear_bwd and as_attn_bwd for 128-aligned feature counts
// (autograd of models/vision_transformer.py:47-59, 75-77, 84; mae_bbox_head_rec.py:148-168).
#include <utility>
#include "common.h"

namespace {

#ifndef AS_TN_ABLATE
#define AS_TN_ABLATE 0                    // (timing experiments on the 128 x 128 kernel: 1 no LDS-DMA in the loop, 2 no barrier, 3 no MFMAs)
#endif
constexpr int TN_T = 128;                 // features per tile side
constexpr int TN_GM = 32;                 // token rows per stage (two k16 steps)
constexpr int TN_NT = 256;
constexpr int TN_TILE_B = TN_GM * TN_T * 2;        // 8 KiB: one operand tile of a stage
constexpr int TN_STAGE_B = 2 * TN_TILE_B;          // dY tile | X tile

typedef __attribute__((ext_vector_type(2))) unsigned tn_u32x2;

__device__ __forceinline__ unsigned tn_lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void tn_dma16(unsigned voff, const char* sbase, unsigned lds_dst) {
  // global_load_lds_dwordx4, saddr form (see sdpa.hip lds_dma16): 64 lanes x 16 B from sbase + voff[lane] -> LDS [m0 + 16 lane]
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int... I, typename F> __device__ __forceinline__ void static_for_tn_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for_tn(F&& f) {
  static_for_tn_impl(std::make_integer_sequence<int, N>{}, f);
}
template <int OFF> __device__ __forceinline__ void tn_read_tr(tn_u32x2& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// grid (tiles, S).  part: fp32 [S][Nout][K].  chunk: token rows per split (multiple of TN_GM).
template <int TN_NSTAGE>
__global__ __launch_bounds__(TN_NT, TN_NSTAGE == 4 ? 2 : 3) void gemm_tn_splitk_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ x,
                                                                  float* __restrict__ part, float* __restrict__ db_part, int M, int Nout,
                                                                  int K, int chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, half = lane >> 5;
  const int nt_k = K / TN_T, tiles = (Nout / TN_T) * nt_k;
  // XCD-aware tile order as in gemm.hip: XCD x walks a contiguous range of tiles, k fastest (a dY tile is fetched into one L2)
  const int per = (tiles + 7) >> 3;
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= tiles) return;
  const int n0 = (tile / nt_k) * TN_T, k0 = (tile % nt_k) * TN_T;
  const int m_lo = blockIdx.y * chunk, m_hi = min(M, m_lo + chunk);
  const int nst = (m_hi - m_lo + TN_GM - 1) / TN_GM;                 // stages of this split (>= 1 by construction)

  // ---- loader: a stage is 8 + 8 one-KiB pieces (4 token rows x 256 B each); wave w moves pieces 2w, 2w+1 of both tiles.
  // lane l of a piece: row l >> 4, LDS chunk position l & 15, which must receive SOURCE chunk (l & 15) ^ ((row & 3) << 2)
  const int ld_row = lane >> 4;                                      // row inside a piece (= row & 3: pieces start at multiples of 4)
  const unsigned ld_col = (unsigned)(((lane & 15) ^ (ld_row << 2)) * 16);
  const char* const dy_b = reinterpret_cast<const char*>(dy) + (size_t)n0 * 2;
  const char* const x_b = reinterpret_cast<const char*>(x) + (size_t)k0 * 2;
  const unsigned smem_base = tn_lds_addr(smem);
  auto stage = [&](int st, int buf) {
    const unsigned base = smem_base + buf * TN_STAGE_B;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = min(m_lo + st * TN_GM + (wave * 2 + j) * 4 + ld_row, M - 1);   // rows past the end re-read the last row (zeroed below)
      tn_dma16((unsigned)m * (unsigned)(Nout * 2) + ld_col, dy_b, base + (wave * 2 + j) * 1024);
      tn_dma16((unsigned)m * (unsigned)(K * 2) + ld_col, x_b, base + TN_TILE_B + (wave * 2 + j) * 1024);
    }
  };

  // ---- fragment addresses.  Feature block fb (32 wide) of a tile, k16 step s, read q: the 16-lane group g = lane >> 4 reads
  // rows 16 s + 8 (g >> 1) + 4 q .. +3, columns 32 fb + 16 (g & 1) .. +15; lane t = lane & 15 supplies row (t >> 2), columns
  // 4 (t & 3) .. +3 of that block.  Chunk position = ((fb ^ (row & 3)) << 2) | (2 (g & 1) + ((t & 3) >> 1)), + 8 bytes if t odd.
  unsigned fA[2], fB[2];
  {
    const int g = lane >> 4, t = lane & 15;
    const int rowl = 8 * (g >> 1) + (t >> 2);                        // + 16 s + 4 q as immediates
    const int lowc = 2 * (g & 1) + ((t & 3) >> 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int fba = wm * 2 + i, fbb = wn * 2 + i;
      fA[i] = smem_base + rowl * 256 + ((((fba ^ (t >> 2)) << 2) | lowc) << 4) + (t & 1) * 8;
      fB[i] = smem_base + TN_TILE_B + rowl * 256 + ((((fbb ^ (t >> 2)) << 2) | lowc) << 4) + (t & 1) * 8;
    }
  }

  f32x16 acc[2][2];                                                  // [i: n block][j: k block], D[k][n]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // bias gradient: db[n] = sum_m dY[m][n] falls out of the dY fragments -- the workgroups of the first k tile (wave column 0)
  // add up the eight token values every fragment holds; partial per token range, summed in range order by the bias blocks of tn_reduce_kernel
  const bool want_db = db_part != nullptr && (tile % nt_k) == 0 && wn == 0;
  float dbacc[2] = {0.0f, 0.0f};
  stage(0, 0);
  if (nst > 1) stage(1, 1);
  if (TN_NSTAGE == 4 && nst > 2) stage(2, 2);
  for (int st0 = 0; st0 < nst; st0 += TN_NSTAGE) {
    static_for_tn<TN_NSTAGE>([&](auto slot_c) {
      constexpr int SLOT = decltype(slot_c)::value;
      const int st = st0 + SLOT;
      if (st >= nst) return;
      // my pieces of stage st have landed when at most the (up to NSTAGE - 2) newer stages are in flight: 4 LDS-DMA per stage
      const int newer = min(TN_NSTAGE - 2, nst - 1 - st);
      if (newer == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (AS_TN_ABLATE != 2) __builtin_amdgcn_s_barrier();           // publishes stage st; everyone is done reading stage st-1
      asm volatile("" ::: "memory");
      if (AS_TN_ABLATE != 1 && st + TN_NSTAGE - 1 < nst) stage(st + TN_NSTAGE - 1, (SLOT + TN_NSTAGE - 1) % TN_NSTAGE);
      // fragments of the stage: [operand][block][k16 step][read]; the first k16 step's eight reads are waited for alone
      // (LDS returns in order), so its MFMAs run while the second step's reads are still in flight
      tn_u32x2 ra[2][2][2], rb[2][2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        tn_read_tr<SLOT * TN_STAGE_B + 0 * 4096 + 0 * 1024>(ra[i][0][0], fA[i]);
        tn_read_tr<SLOT * TN_STAGE_B + 0 * 4096 + 1 * 1024>(ra[i][0][1], fA[i]);
        tn_read_tr<SLOT * TN_STAGE_B + 0 * 4096 + 0 * 1024>(rb[i][0][0], fB[i]);
        tn_read_tr<SLOT * TN_STAGE_B + 0 * 4096 + 1 * 1024>(rb[i][0][1], fB[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        tn_read_tr<SLOT * TN_STAGE_B + 1 * 4096 + 0 * 1024>(ra[i][1][0], fA[i]);
        tn_read_tr<SLOT * TN_STAGE_B + 1 * 4096 + 1 * 1024>(ra[i][1][1], fA[i]);
        tn_read_tr<SLOT * TN_STAGE_B + 1 * 4096 + 0 * 1024>(rb[i][1][0], fB[i]);
        tn_read_tr<SLOT * TN_STAGE_B + 1 * 4096 + 1 * 1024>(rb[i][1][1], fB[i]);
      }
      auto wait_step = [&](auto s_c, auto left_c) {
        constexpr int S_ = decltype(s_c)::value;
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(ra[0][S_][0]), "+v"(ra[0][S_][1]), "+v"(ra[1][S_][0]), "+v"(ra[1][S_][1]), "+v"(rb[0][S_][0]),
                       "+v"(rb[0][S_][1]), "+v"(rb[1][S_][0]), "+v"(rb[1][S_][1])
                     : "n"(decltype(left_c)::value));
        __builtin_amdgcn_sched_barrier(0);
      };
      const int mrow = m_lo + st * TN_GM;                            // first token row of the stage
      const bool ragged = mrow + TN_GM > M;                          // only the last stage of the last split
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 0) wait_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
        else wait_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        Frag<__bf16> fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned wa[4] = {ra[i][s][0][0], ra[i][s][0][1], ra[i][s][1][0], ra[i][s][1][1]};
          const unsigned wb[4] = {rb[i][s][0][0], rb[i][s][0][1], rb[i][s][1][0], rb[i][s][1][1]};
          fa[i].v = *reinterpret_cast<const bf16x8*>(wa);
          fb[i].v = *reinterpret_cast<const bf16x8*>(wb);
          if (ragged) {                                              // token rows >= M were clamped re-reads: contribute zero
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (mrow + 16 * s + 8 * half + e >= M) { fa[i].v[e] = (__bf16)0.0f; fb[i].v[e] = (__bf16)0.0f; }
          }
        }
        if (want_db) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float t = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t += (float)fa[i].v[e];
            dbacc[i] += t;
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (AS_TN_ABLATE == 3) { acc[i][j][0] += (float)fa[i].v[0] + (float)fb[j].v[1]; continue; }
            acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);          // D[k][n]
          }
      }
    });
  }

  if (want_db) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v = dbacc[i] + __shfl_xor(dbacc[i], 32);          // the two halves hold the two 8-token runs of every k16 step
      if (half == 0) db_part[(size_t)blockIdx.y * Nout + n0 + wm * 64 + i * 32 + li] = v;
    }
  }
  // ---- fp32 partial of this token range: lane = row n of dW, registers 4g .. 4g+3 = 4 consecutive columns k
  float* out = part + (size_t)blockIdx.y * Nout * K;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wm * 64 + i * 32 + li;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = k0 + wn * 64 + j * 32 + 8 * g + 4 * half;
        *reinterpret_cast<float4*>(out + (size_t)n * K + k) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
  }
}

// ---- 256 (n) x 128 (k) tiles: four waves of 128 x 64 -------------------------------------------------------------
// The 128 x 128 kernel above issues two transposing LDS reads per MFMA and stages 16 KiB per 64 MFMAs: with three workgroups
// per CU its LDS pipe is asked for more cycles than its matrix pipe (reads 4 waves x 16 x 2 cycles + ~200 cycles of LDS-DMA
// writes against 256 MFMA cycles per stage and SIMD) -- it waits half of its wave cycles.  Here a wave owns 128 rows of dW x
// 64 columns: 24 reads per 16 MFMAs and 24 KiB staged per 128, the same kernel otherwise (32-token stages, 3-deep LDS-DMA
// ring, fixed-order fp32 partials per token range, bias gradient from the dY fragments).  The fragment reads go through the
// compiler's transposing-read builtin (hipcc tracks their lgkmcnt), issued for the whole stage before its first MFMA.
// dY tile: 32 token rows x 512 B, X tile: 32 x 256 B; 16-byte chunk c of token row r at position c ^ ((r & 3) << 2).
constexpr int TW_N = 256, TW_K = 128;
constexpr int TW_A_B = TN_GM * TW_N * 2, TW_B_B = TN_GM * TW_K * 2, TW_STAGE_B = TW_A_B + TW_B_B;   // 16 + 8 KiB
constexpr int TW_NSTAGE = 3;
typedef short tw_i16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const char* tw_lds_ptr;
__device__ __forceinline__ void tw_frag(Frag<__bf16>& f, tw_lds_ptr p0, tw_lds_ptr p1) {
  const tw_i16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tw_i16x4*)p0);
  const tw_i16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tw_i16x4*)p1);
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  const i16x8 w = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  f.v = __builtin_bit_cast(bf16x8, w);
}

__global__ __launch_bounds__(TN_NT, 2) void gemm_tn_wide_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ x,
                                                              float* __restrict__ part, float* __restrict__ db_part, int M, int Nout,
                                                              int K, int chunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, half = lane >> 5;
  const int nt_k = K / TW_K, tiles = (Nout / TW_N) * nt_k;
  const int per = (tiles + 7) >> 3;                                  // XCD-aware tile order, k fastest (a dY tile stays in one L2)
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= tiles) return;
  const int n0 = (tile / nt_k) * TW_N, k0 = (tile % nt_k) * TW_K;
  const int m_lo = blockIdx.y * chunk, m_hi = min(M, m_lo + chunk);
  const int nst = (m_hi - m_lo + TN_GM - 1) / TN_GM;

  // ---- loader: a stage is 16 one-KiB pieces of dY (2 token rows x 512 B) and 8 of X (4 rows x 256 B); wave w moves dY pieces
  // 4w .. 4w+3 and X pieces 2w, 2w+1.  LDS is written lane-linear, so the swizzle is applied to the SOURCE chunk.
  const int a_row = lane >> 5, b_row = lane >> 4;                    // row inside a piece
  const char* const dy_b = reinterpret_cast<const char*>(dy) + (size_t)n0 * 2;
  const char* const x_b = reinterpret_cast<const char*>(x) + (size_t)k0 * 2;
  const unsigned smem_base = tn_lds_addr(smem);
  auto stage = [&](int st, int buf) {
    const unsigned base = smem_base + buf * TW_STAGE_B;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = (wave * 4 + j) * 2 + a_row;                      // token row of the stage
      const int m = min(m_lo + st * TN_GM + r, M - 1);               // rows past the end re-read the last row (zeroed below)
      tn_dma16((unsigned)m * (unsigned)(Nout * 2) + (unsigned)(((lane & 31) ^ ((r & 3) << 2)) * 16), dy_b, base + (wave * 4 + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 4 + b_row;
      const int m = min(m_lo + st * TN_GM + r, M - 1);
      tn_dma16((unsigned)m * (unsigned)(K * 2) + (unsigned)(((lane & 15) ^ ((r & 3) << 2)) * 16), x_b, base + TW_A_B + (wave * 2 + j) * 1024);
    }
  };

  // ---- fragment pointers (slot 0, k16 step 0, read 0): the 16-lane group g = lane >> 4 reads rows 8 (g >> 1) + (t >> 2) [+ 16 s
  // + 4 q as immediates], columns 32 blk + 16 (g & 1) + 4 (t & 3) .. +3 of the block; chunk = 4 blk' + 2 (g & 1) + ((t & 3) >> 1)
  tw_lds_ptr pa[4], pb[2];
  {
    const int g = lane >> 4, t = lane & 15;
    const int rowl = 8 * (g >> 1) + (t >> 2), lowc = 2 * (g & 1) + ((t & 3) >> 1), sw = (t >> 2) << 2, sub = (t & 1) * 8;
    const tw_lds_ptr base = (tw_lds_ptr)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) pa[i] = base + (rowl * 512 + ((((wm * 4 + i) * 4 + lowc) ^ sw) << 4) + sub);
#pragma unroll
    for (int j = 0; j < 2; ++j) pb[j] = base + (TW_A_B + rowl * 256 + ((((wn * 2 + j) * 4 + lowc) ^ sw) << 4) + sub);
  }

  f32x16 acc[4][2];                                                  // [i: n block][j: k block], D[k][n]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const bool want_db = db_part != nullptr && (tile % nt_k) == 0 && wn == 0;
  float dbacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};

  stage(0, 0);
  if (nst > 1) stage(1, 1);
  for (int st0 = 0; st0 < nst; st0 += TW_NSTAGE) {
    static_for_tn<TW_NSTAGE>([&](auto slot_c) {
      constexpr int SLOT = decltype(slot_c)::value;
      const int st = st0 + SLOT;
      if (st >= nst) return;
      // my pieces of stage st have landed when at most the one newer stage is in flight: 6 LDS-DMA per wave and stage
      if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                  // publishes stage st; everyone is done reading stage st-1
      asm volatile("" ::: "memory");
      if (st + TW_NSTAGE - 1 < nst) stage(st + TW_NSTAGE - 1, (SLOT + TW_NSTAGE - 1) % TW_NSTAGE);
      const int mrow = m_lo + st * TN_GM;
      const bool ragged = mrow + TN_GM > M;                          // only the last stage of the last split
      Frag<__bf16> fa[2][4], fb[2][2];                               // [k16 step][block]
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        constexpr int O = SLOT * TW_STAGE_B;
#pragma unroll
        for (int i = 0; i < 4; ++i) tw_frag(fa[sk][i], pa[i] + (O + (16 * sk) * 512), pa[i] + (O + (16 * sk + 4) * 512));
#pragma unroll
        for (int j = 0; j < 2; ++j) tw_frag(fb[sk][j], pb[j] + (O + (16 * sk) * 256), pb[j] + (O + (16 * sk + 4) * 256));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        if (ragged) {                                                // token rows >= M were clamped re-reads: contribute zero
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (mrow + 16 * sk + 8 * half + e >= M) {
#pragma unroll
              for (int i = 0; i < 4; ++i) fa[sk][i].v[e] = (__bf16)0.0f;
#pragma unroll
              for (int j = 0; j < 2; ++j) fb[sk][j].v[e] = (__bf16)0.0f;
            }
        }
        if (want_db) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float t = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t += (float)fa[sk][i].v[e];
            dbacc[i] += t;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fb[sk][j], fa[sk][i], acc[i][j]);          // D[k][n]
      }
    });
  }

  if (want_db) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = dbacc[i] + __shfl_xor(dbacc[i], 32);          // the two halves hold the two 8-token runs of every k16 step
      if (half == 0) db_part[(size_t)blockIdx.y * Nout + n0 + wm * 128 + i * 32 + li] = v;
    }
  }
  // ---- fp32 partial of this token range: lane = row n of dW, registers 4g .. 4g+3 = 4 consecutive columns k
  float* out = part + (size_t)blockIdx.y * Nout * K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wm * 128 + i * 32 + li;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = k0 + wn * 64 + j * 32 + 8 * g + 4 * half;
        *reinterpret_cast<float4*>(out + (size_t)n * K + k) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
  }
}

// The last `db_blocks` workgroups of the grid sum the bias-gradient partials instead ([S][C] -> [C]; same fixed order): one
// launch for both reductions (the separate 5 us bias launch was 83 launches = 0.47 ms of the training step).
template <typename T>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, T* __restrict__ out, size_t n4, int S,
                                                        size_t stride, const float* __restrict__ db_part, float* __restrict__ db,
                                                        int C, int db_blocks) {
  const int main_blocks = (int)gridDim.x - db_blocks;
  if ((int)blockIdx.x >= main_blocks) {
    const int c = ((int)blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (c >= C) return;
    float sum = 0.0f;
    for (int i = 0; i < S; ++i) sum += db_part[(size_t)i * C + c];
    db[c] = sum;
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)main_blocks * blockDim.x) {
    float4 a = *reinterpret_cast<const float4*>(part + i * 4);
    for (int sp = 1; sp < S; ++sp) {
      const float4 b = *reinterpret_cast<const float4*>(part + sp * stride + i * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if constexpr (sizeof(T) == 2) {
      bf16x4 v = {(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w};
      *reinterpret_cast<bf16x4*>(out + i * 4) = v;
    } else {
      *reinterpret_cast<float4*>(out + i * 4) = a;
    }
  }
}

// fp32 column sums of a bf16 matrix [R, C] (C % 8 == 0) with 16-byte loads: thread = 8 columns x every 8th row of a slice,
// slices -> partials [TN_CS_SLICES][C], then one fixed-order pass.  (The bias gradient, when no transposed copy exists.)
constexpr int TN_CS_SLICES = 32;
__global__ __launch_bounds__(256) void tn_colsum_partial_kernel(const __bf16* __restrict__ in, float* __restrict__ part, int R, int C) {
  __shared__ float sm[8][32][9];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cx) * 8;
  const int rows = (R + TN_CS_SLICES - 1) / TN_CS_SLICES;
  const int r0 = blockIdx.y * rows, r1 = min(R, r0 + rows);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C)
    for (int r = r0 + ry; r < r1; r += 8) {
      const uint4 d = *reinterpret_cast<const uint4*>(in + (size_t)r * C + c);
      const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[2 * j] += __uint_as_float(w[j] << 16); s[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u); }
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) sm[ry][cx][j] = s[j];
  __syncthreads();
  if (ry == 0 && c < C) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = sm[0][cx][j];
#pragma unroll
      for (int y = 1; y < 8; ++y) t += sm[y][cx][j];
      part[(size_t)blockIdx.y * C + c + j] = t;
    }
  }
}
__global__ __launch_bounds__(256) void tn_colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.0f;
  for (int i = 0; i < TN_CS_SLICES; ++i) s += part[(size_t)i * C + c];
  out[c] = s;
}

}  // namespace

// ---- internal entry points (used by attn_bwd.hip; not part of the C ABI) ----
bool as_tn_applies(int M, int Nout, int K) {
  static const bool off = getenv("AS_BWD_TRANSPOSED") != nullptr;       // A/B switch: the round-3 path with transposed copies
  return !off && M >= TN_GM && Nout % TN_T == 0 && K % TN_T == 0 && (size_t)M * Nout * 2 < (1ull << 32) &&
         (size_t)M * K * 2 < (1ull << 32);
}
// token ranges: one or two workgroups per CU in all (<= 32 ranges), a multiple of the stage, >= 256 rows
// 256 x 128 tiles (gemm_tn_wide_kernel) when the output has rows for them and at least 64 such tiles -- the MLP's weight
// gradients (72 tiles at ViT-B: fc1 / fc2 dW + db 69.5 / 73.2 -> 61-62 / 66-68 us with two workgroups per CU = 7 token ranges;
// with one per CU it LOSES, 88-90 us: four waves per CU do not hide the per-stage barrier).  The 54 / 18 tiles of the QKV / proj
// gradients and the heads' 8 need more token ranges than their fp32 partials are worth (module backward 627 us either way).
// AS_TN_WIDE=0: always the 128 x 128 kernel
static bool tn_wide(int Nout, int K) {
  static const bool off = getenv("AS_TN_WIDE") != nullptr && atoi(getenv("AS_TN_WIDE")) == 0;
  static const int min_tiles = getenv("AS_TN_WIDE_MIN") ? atoi(getenv("AS_TN_WIDE_MIN")) : 64;
  return !off && Nout % TW_N == 0 && K % TW_K == 0 && (Nout / TW_N) * (K / TW_K) >= min_tiles;
}
static int tn_plan(int M, int Nout, int K, int* splits) {
  if (tn_wide(Nout, K)) {                            // two workgroups per CU: tiles x ranges ~ 512, <= 8 ranges
    const int tiles = (Nout / TW_N) * (K / TW_K);
    static const int slots_env = [] { const char* e = getenv("AS_TN_WIDE_SLOTS"); return e ? atoi(e) : 0; }();   // (experiments)
    int S = (slots_env > 0 ? slots_env : 512) / tiles;
    if (S > 8) S = 8;
    if (S < 1) S = 1;
    int chunk = as_round_up(as_ceil_div(M, S), TN_GM);
    if (chunk < 256) chunk = 256;
    if (chunk > as_round_up(M, TN_GM)) chunk = as_round_up(M, TN_GM);
    *splits = as_ceil_div(M, chunk);
    return chunk;
  }
  const int tiles = (Nout / TN_T) * (K / TN_T);
  static const int slots_env = [] { const char* e = getenv("AS_TN_SLOTS"); return e ? atoi(e) : 0; }();   // (experiments)
  // measured (tools/experiments/dw_tn_bench.py, stages x slots sweep): the MLP's 144 output tiles want two workgroups per CU
  // (fc1 / fc2 dW + db: 123 -> 74 us), a few dozen tiles one (proj 768 x 768: 51 -> 44 us; the heads' 16 tiles: 134 -> 124)
  const int slots = slots_env > 0 ? slots_env : (tiles >= 96 ? 512 : 256);
  int S = slots / (tiles > 0 ? tiles : 1);
  if (S > 32) S = 32;
  if (S < 1) S = 1;
  int chunk = as_round_up(as_ceil_div(M, S), TN_GM);
  if (chunk < 256) chunk = 256;
  if (chunk > as_round_up(M, TN_GM)) chunk = as_round_up(M, TN_GM);
  *splits = as_ceil_div(M, chunk);
  return chunk;
}
size_t as_tn_workspace_bytes(int M, int Nout, int K) {
  int S = 1;
  (void)tn_plan(M, Nout, K, &S);
  return (size_t)S * Nout * K * sizeof(float);
}
size_t as_tn_colsum_workspace_bytes(int C) { return (size_t)TN_CS_SLICES * C * sizeof(float); }

int as_tn_dw(const void* dy, const void* x, void* dW, float* db, float* db_part, int M, int Nout, int K, int dw_f32, void* ws,
             size_t ws_bytes, hipStream_t s) {
  int S = 1;
  const int chunk = tn_plan(M, Nout, K, &S);
  AS_REQUIRE(ws && ws_bytes >= (size_t)S * Nout * K * sizeof(float), AS_E_BADARG, "tn dW: workspace too small");
  AS_REQUIRE(!db || db_part, AS_E_BADARG, "tn dW: db needs its partial buffer ([<= 32][Nout] floats)");
  if (tn_wide(Nout, K)) {
    const int tiles = (Nout / TW_N) * (K / TW_K);
    static std::atomic<bool> attr_w{false};
    if (!attr_w) {
      (void)hipFuncSetAttribute((const void*)gemm_tn_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TW_NSTAGE * TW_STAGE_B);
      attr_w = true;
    }
    hipLaunchKernelGGL(gemm_tn_wide_kernel, dim3(8 * as_ceil_div(tiles, 8), S), dim3(TN_NT), (size_t)TW_NSTAGE * TW_STAGE_B, s,
                       (const __bf16*)dy, (const __bf16*)x, (float*)ws, db ? db_part : nullptr, M, Nout, K, chunk);
    AS_CHECK_LAUNCH("gemm_tn_wide");
  } else {
  const int tiles = (Nout / TN_T) * (K / TN_T);
  static const int stages = [] { const char* e = getenv("AS_TN_STAGES"); return e && atoi(e) == 4 ? 4 : 3; }();   // (experiments: 4-deep ring, 2 workgroups per CU)
  const size_t lds = (size_t)stages * TN_STAGE_B;
  static std::atomic<bool> attr{false};
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_splitk_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TN_STAGE_B);
    (void)hipFuncSetAttribute((const void*)gemm_tn_splitk_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * TN_STAGE_B);
    attr = true;
  }
  if (stages == 3)
    hipLaunchKernelGGL(gemm_tn_splitk_kernel<3>, dim3(8 * as_ceil_div(tiles, 8), S), dim3(TN_NT), lds, s, (const __bf16*)dy,
                       (const __bf16*)x, (float*)ws, db ? db_part : nullptr, M, Nout, K, chunk);
  else
    hipLaunchKernelGGL(gemm_tn_splitk_kernel<4>, dim3(8 * as_ceil_div(tiles, 8), S), dim3(TN_NT), lds, s, (const __bf16*)dy,
                       (const __bf16*)x, (float*)ws, db ? db_part : nullptr, M, Nout, K, chunk);
  AS_CHECK_LAUNCH("gemm_tn_splitk");
  }
  const size_t n4 = (size_t)Nout * K / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  const int db_blocks = db ? as_ceil_div(Nout, 256) : 0;
  if (dw_f32)
    hipLaunchKernelGGL(tn_reduce_kernel<float>, dim3(grid + db_blocks), dim3(256), 0, s, (const float*)ws, (float*)dW, n4, S,
                       (size_t)Nout * K, (const float*)db_part, db, Nout, db_blocks);
  else
    hipLaunchKernelGGL(tn_reduce_kernel<__bf16>, dim3(grid + db_blocks), dim3(256), 0, s, (const float*)ws, (__bf16*)dW, n4, S,
                       (size_t)Nout * K, (const float*)db_part, db, Nout, db_blocks);
  AS_CHECK_LAUNCH("tn_reduce");
  return AS_OK;
}

int as_tn_colsum(const void* g, float* out, float* part, int R, int C, hipStream_t s) {
  hipLaunchKernelGGL(tn_colsum_partial_kernel, dim3(as_ceil_div(C, 256), TN_CS_SLICES), dim3(256), 0, s, (const __bf16*)g, part, R, C);
  AS_CHECK_LAUNCH("tn_colsum_partial");
  hipLaunchKernelGGL(tn_colsum_final_kernel, dim3(as_ceil_div(C, 256)), dim3(256), 0, s, (const float*)part, out, C);
  AS_CHECK_LAUNCH("tn_colsum_final");
  return AS_OK;
}
