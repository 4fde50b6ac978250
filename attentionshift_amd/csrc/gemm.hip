// Linear / QKV projection GEMM for gfx950:  C[M,Nout] = act(A[M,K] . W[Nout,K]^T + bias)
//
// Replaces nn.Linear in Attention / Mlp (reference models/vision_transformer.py:47-59, 75-77, 84).
// 128x128x32 workgroup tile, 4 waves (2x2), each wave 2x2 MFMA 32x32 tiles (64 accumulator VGPRs);
// both operands are K-contiguous so A and W fragments are 16-byte (bf16) row reads; LDS rows are padded
// by 16 B so the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots (conflict-free).
// The QKV epilogue writes k as [B,h,Npad,64], q in the fragment-major tile layout of common.h (qf_elem) and V
// TRANSPOSED as [B,h,64,Npad]; V tiles are computed with swapped MFMA operands so lanes run along tokens and
// the transposed store stays coalesced.
#include <utility>
#include "common.h"
#include "gemm_epi.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;

template <typename T> struct GemmCfg {
  static constexpr int ROW_BYTES = BK * (int)sizeof(T);
  static constexpr int PITCH = ROW_BYTES + 16;            // bytes
  static constexpr int CHUNKS_PER_ROW = ROW_BYTES / 16;   // 16-byte chunks
  static constexpr int CHUNKS = BM * CHUNKS_PER_ROW;      // per operand tile
  static constexpr int PER_THREAD = CHUNKS / NT;
  static constexpr int FRAG_BYTES = 8 * (int)sizeof(T);   // one lane's k16-step fragment
};

struct QkvEpi {
  void* q; void* k; void* vt;
  int N, Npad, D, h;
  int stagger;     // two-workgroups-per-CU tiles: the second wave of workgroups starts this many s_sleep(127) late (0 = off)
};

// accumulators -> memory.  MODE 0: row-major out (+bias, +GELU).  MODE 1: q,k [B,h,Npad,64] and V^T [B,h,64,Npad].
template <typename T, int MODE>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[2][2], const bool (&swapped)[2], const float* __restrict__ bias,
                                              T* __restrict__ out, int M, int Nout, int act, const QkvEpi& epi, int m0,
                                              int n0, int wm, int wn, int li, int half) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rbase = m0 + wm * 64 + i * 32, cbase = n0 + wn * 64 + j * 32;
      if (MODE == 0) {
        const int col = cbase + li;
        const float bv = (bias != nullptr && col < Nout) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + acc_row(r, half);
          if (row < M && col < Nout) {
            float v = acc[i][j][r] + bv;
            if (act == 1) v = gelu_exact(v);
            else if (act == 4) v = fmaxf(v, 0.0f);       // ReLU (the point head's FFN, visual_transformer_det.py:36)
            out[(size_t)row * Nout + col] = from_f32<T>(v);
          }
        }
      } else {
        const int which = cbase / epi.D;                       // 0 q, 1 k, 2 v (uniform per block)
        const int head = (cbase % epi.D) / 64, dd0 = cbase % 64;
        if (!swapped[j]) {
          const int col = cbase + li, dd = dd0 + li;
          const float bv = (bias != nullptr && col < Nout) ? bias[col] : 0.0f;
          T* dst = reinterpret_cast<T*>(which == 0 ? epi.q : epi.k);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + acc_row(r, half);
            if (row < M && col < Nout) {
              const int b = row / epi.N, n = row - b * epi.N;
              const size_t at = which == 0 ? qf_elem((size_t)(b * epi.h + head), epi.Npad, n, dd)   // fragment-major q
                                           : ((size_t)(b * epi.h + head) * epi.Npad + n) * 64 + dd;
              dst[at] = from_f32<T>(which == 0 ? (acc[i][j][r] + bv) * AS_QSCALE : acc[i][j][r] + bv);   // q pre-scaled
            }
          }
        } else {
          // D[i = feature][j = token]: lanes run along tokens, registers along features
          const int row = rbase + li;                          // token
          T* dst = reinterpret_cast<T*>(epi.vt);
          if (row < M) {
            const int b = row / epi.N, n = row - b * epi.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int f = acc_row(r, half);
              const int col = cbase + f;
              if (col < Nout) {
                const float bv = bias != nullptr ? bias[col] : 0.0f;
                dst[((size_t)(b * epi.h + head) * 64 + dd0 + f) * epi.Npad + n] = from_f32<T>(acc[i][j][r] + bv);
              }
            }
          }
        }
      }
    }
}

template <typename T, int MODE>   // MODE 0: plain row-major out (+act)   MODE 1: qkv scatter
__global__ __launch_bounds__(NT) void gemm_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                  const float* __restrict__ bias, T* __restrict__ out,
                                                  int M, int Nout, int K, int act, QkvEpi epi) {
  using Cfg = GemmCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Bs = smem + BM * Cfg::PITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  // staging assignment: chunk c -> (row = c / CHUNKS_PER_ROW, col16 = c % CHUNKS_PER_ROW)
  uint4 ra[Cfg::PER_THREAD], rb[Cfg::PER_THREAD];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < Cfg::PER_THREAD; ++i) {
      const int c = tid + i * NT;
      const int row = c / Cfg::CHUNKS_PER_ROW, ch = c % Cfg::CHUNKS_PER_ROW;
      const int ar = min(m0 + row, M - 1), br = min(n0 + row, Nout - 1);
      const char* pa = reinterpret_cast<const char*>(A + (size_t)ar * K + (size_t)kt * BK) + ch * 16;
      const char* pb = reinterpret_cast<const char*>(W + (size_t)br * K + (size_t)kt * BK) + ch * 16;
      ra[i] = *reinterpret_cast<const uint4*>(pa);
      rb[i] = *reinterpret_cast<const uint4*>(pb);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < Cfg::PER_THREAD; ++i) {
      const int c = tid + i * NT;
      const int row = c / Cfg::CHUNKS_PER_ROW, ch = c % Cfg::CHUNKS_PER_ROW;
      *reinterpret_cast<uint4*>(As + row * Cfg::PITCH + ch * 16) = ra[i];
      *reinterpret_cast<uint4*>(Bs + row * Cfg::PITCH + ch * 16) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // V tiles (qkv mode) use swapped operands: wave-uniform per 32-column block
  bool swapped[2] = {false, false};
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) swapped[j] = (n0 + wn * 64 + j * 32) >= 2 * epi.D;
  }

  const int nk = K / BK;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + li;
        fa[i].load16B(reinterpret_cast<const T*>(As + row * Cfg::PITCH + (ks * 16 + half * 8) * (int)sizeof(T)));
        const int col = wn * 64 + i * 32 + li;
        fb[i].load16B(reinterpret_cast<const T*>(Bs + col * Cfg::PITCH + (ks * 16 + half * 8) * (int)sizeof(T)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (MODE == 1 && swapped[j]) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
          else acc[i][j] = mma32(fa[i], fb[j], acc[i][j]);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }

  gemm_epilogue<T, MODE>(acc, swapped, bias, out, M, Nout, act, epi, m0, n0, wm, wn, li, half);
}

// ---------------------------------------------------------------------------------------------------------
// bf16 fast path: 128x128 output tile, K step 32, operands streamed global -> LDS with global_load_lds_dwordx4 (no
// VGPR round trip) into a 3-deep LDS ring (3 x 16 KiB, three workgroups per CU): two K steps are in flight while one is consumed, with
// counted `s_waitcnt vmcnt(N)` + raw s_barrier so the LDS-DMA spans barriers (ONE barrier per K step; __syncthreads
// would drain vmcnt(0) and expose a full L2/HBM latency per step -- the short-K GEMMs of this path, K = 768, have only
// 24 steps to hide it in).  The LDS image of a tile is lane-linear ([row][4 x 16-B chunks], 1 KiB per wave
// instruction), so the bank-conflict swizzle is applied on the SOURCE address: LDS chunk c of row r holds global chunk
// c ^ ((r >> 2) & 3) -- four 64-B rows share one 256-B bank row, and with that key the 16-lane groups ds_read_b128 is
// serviced in ({0-3,12-15,20-27}, ...) hit 16 distinct slots.  Fragment reads are issued in inline asm (hipcc would
// otherwise wait vmcnt(0) before any LDS read that follows an LDS-DMA).
// All MFMAs are issued as D[n][m] = W . A^T, so a lane owns ONE output row (token) m and its accumulator registers run
// along the output columns n in groups of 4: the tile is staged through LDS (reusing the ring) with 8-byte writes and
// leaves as 16-byte fully coalesced global stores (row-major out, k, fragment-major q); V^T tiles are stored straight
// from the accumulators (lanes run along tokens: 128 contiguous bytes per store instruction).
// ---------------------------------------------------------------------------------------------------------
#ifndef AS_GEMM_ABLATE
#define AS_GEMM_ABLATE 0                     // timing ablations (tools/experiments/gemm_ablate.py); results are wrong when != 0
#endif
constexpr int GK = 32;                       // K step (elements)
#ifndef AS_GEMM_EPI_PIPE
#define AS_GEMM_EPI_PIPE 0                   // 1 = staged epilogue in RI slices: stage slice i, barrier, issue its stores, convert slice
#endif                                       // i + 1 under them.  Measured round 5 (same box, us, 0 / 1): QKV-shaped 48.5-49.6 / 48.3-49.3,
                                             // fc1 57.1-57.8 / 56.0-57.6, fc1 + GELU 67.5-68.5 / 67.6-67.7, fc2 50.2 / 48.7-50.0, as_qkv_fwd
                                             // 53.6-56.9 / 54.2-58.7: no difference -- the stores were not what the epilogue waits for
#ifndef AS_GEMM_VT_STAGED
#define AS_GEMM_VT_STAGED 0                  // QKV epilogue: 1 = V^T tiles transposed through LDS and stored 16 bytes at a time.  Measured
                                             // round 5 (as_qkv_fwd, config 2, same box, bitwise equal): 56.0-60.8 us against 53.9-57.5 us
                                             // for the two-byte stores straight from the accumulators (0): the stores are fire-and-forget,
                                             // the staging adds two barriers and 64 ds_write_b16 per lane -- kept for the record
#endif
#ifndef AS_GEMM_K32_STAGES
#define AS_GEMM_K32_STAGES 3                 // ring depth of the K-step-32 tiles: 2 K steps in flight + 1 consumed
#endif
constexpr int G_NSTAGE = AS_GEMM_K32_STAGES;
// The kernel is built for two tile heights, WM = wave rows of 64 tokens: WM = 2 -> 128 x 128 tile, 256 threads, 48 KiB
// ring, 3 workgroups per CU; WM = 4 -> 256 x 128 tile, 512 threads, 72 KiB ring, 2 workgroups per CU.  The per-wave code
// (64 x 64 outputs, 8 MFMAs per K step) is the same; the tall tile moves 3/4 of the operand bytes per flop through the
// L2 -> LDS path (which is what bounds this kernel: tools/experiments/gemm_ablate.py), the short one balances better
// when there are few tiles.
// WN = wave columns of 64 outputs (2 -> 128 output columns, 4 -> 256), RI = 32-row blocks per wave (2 -> 64 rows, 4 -> 128:
// 6 fragment reads per 8 MFMAs instead of 4 per 4, accumulators 128 registers).
// KS = k16 MFMA steps per stage: KS = 2 -> K step 32 (a 64-byte row segment per stage: every LDS-DMA instruction touches
// 16 HALF cache lines, and the other half of each 128-byte line is requested again by the next stage); KS = 4 -> K step 64:
// whole 128-byte lines per request, half the barriers; two stages of 64 KiB for the 256 x 256 tile.
#ifndef AS_GEMM_K64_STAGES
#define AS_GEMM_K64_STAGES 2
#endif
template <int WM, int WN = 2, int RI = 2, int KS = 2, int NJ = 2> struct GTile {
  static constexpr int BM_ = 32 * RI * WM, BN_ = 32 * NJ * WN, NT_ = 64 * WM * WN;   // NJ: 32-column blocks per wave
  static constexpr int GK_ = 16 * KS, ROWB = 2 * GK_;        // K elements / bytes of one tile row per stage
  static constexpr int PROWS = 1024 / ROWB;                  // tile rows per 1-KiB LDS-DMA piece: 16 / 8
  static constexpr int A_BYTES = BM_ * ROWB, W_BYTES = BN_ * ROWB;
  static constexpr int STAGE = A_BYTES + W_BYTES;            // A tile | W tile
  static constexpr int A_PIECES = (BM_ / PROWS) / (WM * WN); // 1-KiB pieces per wave per stage: 2 / 2 / 1 at KS = 2
  static constexpr int W_PIECES = (BN_ / PROWS) / (WM * WN); // 2 / 1 / 1
  static constexpr int LOADS = A_PIECES + W_PIECES;          // LDS-DMA instructions per wave per stage (4 / 3 / 2)
  static constexpr int EPI_PITCH = BN_ * 2 + 16;             // bytes per staged output row
  static constexpr int EPI_BYTES = BM_ * EPI_PITCH + BN_ * 4;
  static constexpr int NSTAGE = KS == 4 ? AS_GEMM_K64_STAGES : G_NSTAGE;
  static constexpr int LDS = NSTAGE * STAGE > EPI_BYTES ? NSTAGE * STAGE : EPI_BYTES;
};

typedef __attribute__((ext_vector_type(4))) unsigned g_u32x4;
__device__ __forceinline__ unsigned g_lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ void g_lds_read128(g_u32x4& dst, unsigned addr) {
  // the immediate is 16 bits; the 96 KiB ring of the 256 x 256 tile crosses it in its last stage (one v_add there)
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr + (unsigned)(OFF & ~0xFFFF)), "n"(OFF & 0xFFFF));
}
template <int LEFT> __device__ __forceinline__ void g_lds_wait4(g_u32x4& a, g_u32x4& b, g_u32x4& c, g_u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(LEFT));   // LDS returns in order
  __builtin_amdgcn_sched_barrier(0);
}
template <int LEFT, int N> __device__ __forceinline__ void g_lds_waitn(g_u32x4* f) {
  static_assert(N == 6 || N == 8, "128 x 64 wave tile: 4 + 2 fragments per k16 step; 128 x 128: 4 + 4");
  if constexpr (N == 6)
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "n"(LEFT));
  else
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "n"(LEFT));
  __builtin_amdgcn_sched_barrier(0);
}
template <int... I, typename F> __device__ __forceinline__ void g_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void g_static_for(F&& f) {
  g_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// swizzle of the 16-byte chunk index inside a tile row, so that each ds_read_b128 lane group (16 lanes: rows {0-3, 12-15,
// 20-27} or {4-11, 16-19, 28-31} of a 32-row block, one chunk column) covers the 16 slots of a 256-byte bank row once:
// 64-byte rows (4 chunks): chunk ^ ((row >> 2) & 3); 128-byte rows (8 chunks): chunk ^ (((row >> 1) & 3) | ((row >> 4) & 1) << 2)
template <int KS> __device__ __forceinline__ int g_swz(int r) {
  return KS == 2 ? ((r >> 2) & 3) : (((r >> 1) & 3) | (((r >> 4) & 1) << 2));
}

template <int MODE, int WM, int WN, int RI, int KS, int NJ = 2>
__global__ __launch_bounds__(64 * WM * WN, (WN == 4 || KS == 4) ? 1 : (WM == 2 && RI == 2 ? 3 : 2)) void gemm_glds_kernel(
    const __bf16* __restrict__ A, const __bf16* __restrict__ W, const float* __restrict__ bias, __bf16* __restrict__ out,
    int M, int Nout, int K, int act, QkvEpi epi) {
  using GT = GTile<WM, WN, RI, KS, NJ>;
  constexpr int GK = GT::GK_, ROWB = GT::ROWB, PROWS = GT::PROWS, CPR = ROWB / 16;   // CPR: 16-byte chunks per tile row
  constexpr int BM = GT::BM_, BN = GT::BN_, NT = GT::NT_, G_TILE_BYTES = GT::A_BYTES, G_STAGE = GT::STAGE;
  constexpr int G_EPI_PITCH = GT::EPI_PITCH, NSTAGE = GT::NSTAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [3 stages][A tile | W tile]; reused by the epilogue
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = __builtin_amdgcn_readfirstlane(wave / WN), wn = __builtin_amdgcn_readfirstlane(wave % WN);
  const int li = lane & 31, half = lane >> 5;
  // XCD-aware tile order.  Workgroup ids are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2), so id
  // -> (xcd = id % 8, slot = id / 8) and XCD x walks the CONTIGUOUS range [x * per, (x+1) * per) of tiles, n fastest:
  // an A tile (128 tokens x K) is fetched into one L2 once and reused by all Nout/128 column tiles, instead of being
  // pulled through the fabric by every XCD (measured: 192 MB fetched per QKV launch for 16 MB of operands).
  const int nt_n = (Nout + BN - 1) / BN, tiles = ((M + BM - 1) / BM) * nt_n;
  // MODE 4 / 5 = stream-K (plain / QKV epilogue): a workgroup contracts a contiguous range of the flattened (tile, K step)
  // space -- at most the tail of one tile, whole tiles, the head of another -- so that every workgroup gets the same number
  // of K steps whatever the tile count (launch_gemm_sk).  A tile cut into np > 1 pieces is finished by the piece that draws
  // the LAST ticket: the others leave their fp32 accumulators in the workspace (the sdpa.hip stream-K hand-off: plain stores,
  // one agent-scope release per workgroup, a relaxed counter), so nobody ever waits for a workgroup that is not running.
  constexpr bool SK = MODE >= 4;
  constexpr int EM = MODE == 4 ? 0 : MODE == 5 ? 1 : MODE;           // which epilogue
  // one (tile, K range): kind 0 = the whole contraction; kind 1 = piece `pidx` of `np` (first piece held by rank r_first)
  auto run_tile = [&](const int tile, const int kbeg, const int klen, const int np, const int pidx, const int r_first) {
  const int m0 = (tile / nt_n) * BM, n0 = (tile % nt_n) * BN;

  // loader: per K step wave w moves NAP one-KiB pieces of A (tile rows 16p .. 16p+15 of piece p = NAP*w + j) and NWP of
  // the BN/16 pieces of W (2 + 2 per wave at 4 waves, 2 + 1 at 8, 1 + 1 at 16)
  const int lr = lane / CPR, lc = lane % CPR;
  constexpr int NAP = GT::A_PIECES, NWP = GT::W_PIECES;
  const char* srcA[NAP];
  const char* srcW[NWP];
#pragma unroll
  for (int j = 0; j < NAP; ++j) {
    const int r = (wave * NAP + j) * PROWS + lr;
    srcA[j] = reinterpret_cast<const char*>(A + (size_t)min(m0 + r, M - 1) * K + kbeg) + (lc ^ g_swz<KS>(r)) * 16;
  }
#pragma unroll
  for (int j = 0; j < NWP; ++j) {
    const int r = (wave * NWP + j) * PROWS + lr;
    srcW[j] = reinterpret_cast<const char*>(W + (size_t)min(n0 + r, Nout - 1) * K + kbeg) + (lc ^ g_swz<KS>(r)) * 16;
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * G_STAGE;
#pragma unroll
    for (int j = 0; j < NAP; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + (size_t)kt * GK * 2),
                                       (__attribute__((address_space(3))) void*)(base + (wave * NAP + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NWP; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[j] + (size_t)kt * GK * 2),
                                       (__attribute__((address_space(3))) void*)(base + G_TILE_BYTES + (wave * NWP + j) * 1024),
                                       16, 0, 0);
  };

  f32x16 acc[RI][NJ];                                // [i: 32-row block of m][j: 32-col block of n], D[n][m] orientation
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // per-lane fragment addresses (loop-invariant): [ks] for the A rows of block i = 0 and the W rows of block j = 0;
  // the second block (+32 rows = +2048 B) and the ring stage are immediates
  const unsigned smem_base = g_lds_addr(smem);
  unsigned offA[KS], offW[KS];
  {
    const int ra = wm * (32 * RI) + li, rb = wn * (32 * NJ) + li;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int g = ks * 2 + half;
      offA[ks] = smem_base + ra * ROWB + ((g ^ g_swz<KS>(ra)) << 4);
      offW[ks] = smem_base + G_TILE_BYTES + rb * ROWB + ((g ^ g_swz<KS>(rb)) << 4);
    }
  }

  const int nk = klen / GK;
  {
  stage(0, 0);
#pragma unroll
  for (int p_ = 1; p_ < NSTAGE - 1; ++p_)
    if (nk > p_) stage(p_, p_);
  for (int kt0 = 0; kt0 < nk; kt0 += NSTAGE) {
    g_static_for<NSTAGE>([&](auto slot_c) {
      constexpr int slot = decltype(slot_c)::value;
      const int kt = kt0 + slot;
      if (kt >= nk) return;
      // my pieces of stage kt have landed when at most the (up to NSTAGE-2) newer stages are still in flight
      const int newer = min(NSTAGE - 2, nk - 1 - kt);
      if (AS_GEMM_ABLATE != 1) {
        if (NSTAGE >= 5 && newer >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * GT::LOADS) : "memory");
        else if (NSTAGE >= 4 && newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GT::LOADS) : "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GT::LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (AS_GEMM_ABLATE != 4) __builtin_amdgcn_s_barrier();   // publishes stage kt; everyone is done reading stage kt-1
      if (AS_GEMM_ABLATE != 1 && kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1, (slot + NSTAGE - 1) % NSTAGE);
      constexpr int NF = RI + NJ;                    // fragments per k16 step: RI of A, NJ of W
      g_static_for<KS / 2>([&](auto kp_c) {          // pairs of k16 steps
      constexpr int k0 = 2 * decltype(kp_c)::value;
      g_u32x4 f[2 * NF];                             // [ks][A0 .. A(RI-1), W0, W1]
      if (AS_GEMM_ABLATE == 3) {
#pragma unroll
        for (int x = 0; x < 2 * NF; ++x) asm volatile("" : "=v"(f[x]));
      }
      if (AS_GEMM_ABLATE != 3) g_static_for<2>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        g_static_for<RI>([&](auto i_c) {
          constexpr int i = decltype(i_c)::value;
          g_lds_read128<slot * G_STAGE + i * 32 * ROWB>(f[ks * NF + i], offA[k0 + ks]);
        });
        g_static_for<NJ>([&](auto j_c) {
          constexpr int j = decltype(j_c)::value;
          g_lds_read128<slot * G_STAGE + j * 32 * ROWB>(f[ks * NF + RI + j], offW[k0 + ks]);
        });
      });
      // the first half's MFMAs start as soon as ITS fragments are back; the second half's reads finish under them
      g_static_for<2>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if (AS_GEMM_ABLATE != 3) {
          if constexpr (RI == 2 && NJ == 2) {
            if (ks == 0) g_lds_wait4<4>(f[0], f[1], f[2], f[3]);
            else g_lds_wait4<0>(f[4], f[5], f[6], f[7]);
          } else {
            if (ks == 0) g_lds_waitn<NF, NF>(&f[0]);
            else g_lds_waitn<0, NF>(&f[NF]);
          }
        }
        if (AS_GEMM_ABLATE == 2) return;
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            Frag<__bf16> fa, fb;
            fa.v = *reinterpret_cast<bf16x8*>(&f[ks * NF + i]);
            fb.v = *reinterpret_cast<bf16x8*>(&f[ks * NF + RI + j]);
            acc[i][j] = mma32(fb, fa, acc[i][j]);    // D[n][m]
          }
      });
      });
    });
  }
  }

  if constexpr (SK) {
    if (np > 1) {
      // tile-local fp32 image of a piece: lane = row, registers 4g .. 4g+3 = 4 consecutive columns -> one float4
      float* const parts = reinterpret_cast<float*>(epi.q);
      int* const arrive = reinterpret_cast<int*>(epi.k);
      int* const done = arrive + tiles;
      // piece p of this tile is rank r_first + p: its FIRST segment when p > 0 (slot 0), its last when p == 0 (slot 1)
      auto part_of = [&](int p) { return parts + ((size_t)(r_first + p) * 2 + (p > 0 ? 0 : 1)) * (size_t)(BM * BN); };
      // this lane's first quad; quad (i, j, g) is the compile-time offset i * 32 * BN + j * 32 + 8 * g floats further on
      const int lane_at = (wm * (32 * RI) + li) * BN + wn * (32 * NJ) + 4 * half;
      __syncthreads();                                  // every wave is done with the ring: smem[0..3] is the ticket box
      int* const box = reinterpret_cast<int*>(smem);
      if (tid == 0) box[0] = __hip_atomic_fetch_add(&arrive[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int ticket = __builtin_amdgcn_readfirstlane(box[0]);
      const bool last = ticket == np - 1;
      if (!last || np > 2) {                            // (np == 2: the finisher adds the other piece -- a + b is commutative;
        float* mine = part_of(pidx) + lane_at;          //  np >= 3: it re-reads ALL pieces in piece order, its own included,
#pragma unroll                                          //  so that the sum does not depend on who arrived last)
        for (int i = 0; i < RI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(mine + i * 32 * BN + j * 32 + 8 * g) =
                  make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
      if (!last) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the compiler may drop the fence's own wait here)
          __hip_atomic_fetch_add(&done[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
      }
      // last arriver: the other pieces have drawn their tickets, i.e. they are running -- a bounded wait
      if (tid == 0) {
        while (__hip_atomic_load(&done[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < np - 1) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      if (np > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my own piece is re-read below
      __syncthreads();
      if (np > 2) {
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
      }
      for (int p = 0; p < np; ++p) {
        if (np == 2 && p == pidx) continue;
        const float* src = part_of(p) + lane_at;
#pragma unroll
        for (int i = 0; i < RI; ++i) {                  // one 32-row block at a time: 4 NJ loads in flight, not 4 NJ RI
          float4 v[NJ * 4];
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              v[j * 4 + g] = *reinterpret_cast<const float4*>(src + i * 32 * BN + j * 32 + 8 * g);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              acc[i][j][4 * g] += v[j * 4 + g].x; acc[i][j][4 * g + 1] += v[j * 4 + g].y;
              acc[i][j][4 * g + 2] += v[j * 4 + g].z; acc[i][j][4 * g + 3] += v[j * 4 + g].w;
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // ---------------- epilogue ----------------
  // lane -> token row m = m0 + wm*64 + i*32 + li; register r of block (i,j) -> column n0 + wn*64 + j*32 + acc_row(r,half)
  if constexpr (MODE == 3) {
    // fp32 partial of this K range, straight from the accumulators: lane = row, registers 4g .. 4g+3 = 4 consecutive columns
    float* part = reinterpret_cast<float*>(epi.q) + (size_t)blockIdx.y * M * Nout;
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int row = m0 + wm * (32 * RI) + i * 32 + li;
      if (row >= M) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = n0 + wn * (32 * NJ) + j * 32 + 8 * g + 4 * half;
          float* dst = part + (size_t)row * Nout + col;
          if (col + 4 <= Nout) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          } else {
            for (int x = 0; x < 4; ++x)
              if (col + x < Nout) dst[x] = acc[i][j][4 * g + x];
          }
        }
    }
    return;
  }
  const bool vtile = EM == 1 && n0 >= 2 * epi.D;     // block-uniform: the whole 128-column tile is V
  if (vtile) {
    if (AS_GEMM_ABLATE == 5) return;
    // V^T [B,h,64,Npad]: rows = features, tokens contiguous.  A tile whose tokens all belong to ONE image is transposed
    // through LDS -- [feature][token + shift] with shift = (first token's index in its image) % 8, so that the 16-byte chunks
    // of the image row and of the LDS row coincide -- and leaves as 16-byte stores along the tokens (8 store instructions per
    // thread instead of 64 two-byte ones).  A tile that straddles two images (one per image boundary) takes the direct path.
    constexpr int VT_PITCH = (BM + 8) * 2 + 16;          // bytes per staged feature row (odd multiple of 16: no bank pile-up)
    const int b_first = m0 / epi.N, last_row = min(m0 + BM, M) - 1;
    if (BN * VT_PITCH <= GT::LDS && last_row / epi.N == b_first && (AS_GEMM_VT_STAGED != 0)) {
      const int n_first = m0 - b_first * epi.N, shift = n_first & 7;
      __syncthreads();                                   // every wave is done with the ring
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int tl = wm * (32 * RI) + i * 32 + li + shift;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int fb_ = wn * (32 * NJ) + j * 32;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f_ = fb_ + acc_row(r, half);
            const float bv = (bias != nullptr && n0 + f_ < Nout) ? bias[n0 + f_] : 0.0f;
            *reinterpret_cast<__bf16*>(smem + f_ * VT_PITCH + tl * 2) = (__bf16)(acc[i][j][r] + bv);
          }
        }
      }
      __syncthreads();
      constexpr int NCH = BM / 8 + 1;                    // 16-byte chunks per staged row (the shifted tile spans one more)
      __bf16* const vt_b = reinterpret_cast<__bf16*>(epi.vt);
      const int n_al = n_first - shift;                  // image-local token of staged column 0 (a multiple of 8)
      const int t_hi = last_row - m0;                    // last valid token of the tile
      for (int idx = tid; idx < BN * NCH; idx += NT) {
        const int f_ = idx / NCH, c = idx - f_ * NCH;
        const int col = n0 + f_;
        if (col >= Nout) continue;
        const int head = (col % epi.D) / 64, dd = col % 64;
        __bf16* dst = vt_b + ((size_t)(b_first * epi.h + head) * 64 + dd) * epi.Npad + n_al + 8 * c;
        const int t0 = 8 * c - shift;                    // tile-local token of the chunk's first element
        const char* src = smem + f_ * VT_PITCH + 16 * c;
        if (t0 >= 0 && t0 + 7 <= t_hi) {
          *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
        } else {
          for (int e_ = 0; e_ < 8; ++e_)
            if (t0 + e_ >= 0 && t0 + e_ <= t_hi) dst[e_] = reinterpret_cast<const __bf16*>(src)[e_];
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int row = m0 + wm * (32 * RI) + i * 32 + li;
      if (row < M) {
        const int b = row / epi.N, n = row - b * epi.N;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int cbase = n0 + wn * (32 * NJ) + j * 32;
          const int head = (cbase % epi.D) / 64, dd0 = cbase % 64;
          __bf16* dst = reinterpret_cast<__bf16*>(epi.vt) + ((size_t)(b * epi.h + head) * 64 + dd0) * epi.Npad + n;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int f_ = acc_row(r, half);
            if (cbase + f_ < Nout) {
              const float bv = bias != nullptr ? bias[cbase + f_] : 0.0f;
              dst[(size_t)f_ * epi.Npad] = (__bf16)(acc[i][j][r] + bv);
            }
          }
        }
      }
    }
    return;
  }
  if (AS_GEMM_ABLATE == 7) {                         // timing experiment: no epilogue at all (accumulators kept alive)
    float keep = 0.0f;
#pragma unroll
    for (int i = 0; i < RI; ++i) keep += acc[i][0][0] + acc[i][NJ - 1][15];
    if (keep == 12345.678f) out[0] = (__bf16)keep;
    return;
  }
  __syncthreads();                                   // every wave is done with the ring: reuse it as the staging tile
  float* bias_s = reinterpret_cast<float*>(smem + BM * G_EPI_PITCH);
  if (tid < BN) bias_s[tid] = (bias != nullptr && n0 + tid < Nout) ? bias[n0 + tid] : 0.0f;
  __syncthreads();
  if (AS_GEMM_ABLATE == 8) {                         // timing experiment: no conversion / staging writes (garbage is stored)
    float keep = 0.0f;
#pragma unroll
    for (int i = 0; i < RI; ++i) keep += acc[i][0][0] + acc[i][NJ - 1][15];
    if (keep == 12345.678f) smem[0] = 1;
  }
  // stage(i): this wave's 32-row block i of the tile -> bias, scale / activation, bf16 -> its rows of the staging tile
  auto stage_block = [&](const int i) {
    char* srow = smem + (wm * (32 * RI) + i * 32 + li) * G_EPI_PITCH;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = wn * (32 * NJ) + j * 32 + 8 * g + 4 * half;   // 4 consecutive columns: registers 4g .. 4g+3
        const float4 bv = *reinterpret_cast<const float4*>(bias_s + c0);
        float v0 = acc[i][j][4 * g] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y, v2 = acc[i][j][4 * g + 2] + bv.z,
              v3 = acc[i][j][4 * g + 3] + bv.w;
        if (EM == 1 && n0 + c0 < epi.D) {                // q columns: stored pre-scaled by log2(e)/8 (common.h)
          v0 *= AS_QSCALE; v1 *= AS_QSCALE; v2 *= AS_QSCALE; v3 *= AS_QSCALE;
        }
        if (act == 1) {
#ifdef AS_GEMM_GELU_SCALAR                               // (A/B switch: the one-value-per-instruction form)
          v0 = gelu_bf16(v0); v1 = gelu_bf16(v1); v2 = gelu_bf16(v2); v3 = gelu_bf16(v3);
#else
          const g_f32x2 ga = gelu_bf16_x2(g_f32x2{v0, v1}), gb = gelu_bf16_x2(g_f32x2{v2, v3});
          v0 = ga.x; v1 = ga.y; v2 = gb.x; v3 = gb.y;
#endif
        } else if (act == 4) {
          v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
        }
        bf16x4 pk = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
        *reinterpret_cast<bf16x4*>(srow + c0 * 2) = pk;
      }
  };
  // store_chunk(rr, ch): 16 bytes (8 columns) of staged row rr -> its place in the output layout
  auto store_chunk = [&](const int rr, const int ch) {
    const int row = m0 + rr, col = n0 + ch * 8;
    if (row >= M || col >= Nout) return;
    const uint4 v = *reinterpret_cast<const uint4*>(smem + rr * G_EPI_PITCH + ch * 16);
    if (AS_GEMM_ABLATE == 6) {                        // timing experiment: staged epilogue without its global stores
      if (v.x == 0x12345678u && v.w == 0x9abcdef0u) out[0] = (__bf16)1.0f;
      return;
    }
    if (EM == 2) {
      // 2x2 / stride-2 transposed convolution: GEMM row = input pixel (b*h + i, j) of a grid epi.N wide, column =
      // (di, dj, co) with co < epi.D -> NHWC output pixel (2 (b*h + i) + di, 2 j + dj); a 16-byte chunk never straddles
      const int bi = row / epi.N, j = row - bi * epi.N;
      const int tap = col / epi.D, co = col - tap * epi.D;
      *reinterpret_cast<uint4*>(out + ((size_t)(2 * bi + (tap >> 1)) * (2 * epi.N) + 2 * j + (tap & 1)) * epi.D + co) = v;
    } else if (EM == 0) {
      if (act == 2) {                                  // (Nout % 8 == 0 on this path: as_linear_gelu_fwd)
        *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(epi.q) + (size_t)row * Nout + col) = v;
        *reinterpret_cast<uint4*>(out + (size_t)row * Nout + col) = gelu_chunk(v);
      } else if (act == 3) {
        const uint4 hv = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(epi.q) + (size_t)row * Nout + col);
        *reinterpret_cast<uint4*>(out + (size_t)row * Nout + col) = dgelu_chunk(v, hv);
      } else if (col + 8 <= Nout) {
        *reinterpret_cast<uint4*>(out + (size_t)row * Nout + col) = v;
      } else {
        const __bf16* e = reinterpret_cast<const __bf16*>(smem + rr * G_EPI_PITCH + ch * 16);
        for (int x = 0; x < Nout - col; ++x) out[(size_t)row * Nout + col + x] = e[x];
      }
    } else {
      const int which = col / epi.D;                   // 0 q, 1 k
      const int head = (col % epi.D) / 64, d0 = col % 64;
      const int b = row / epi.N, n = row - b * epi.N;
      const size_t bh = (size_t)(b * epi.h + head);
      if (which == 0)
        *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(epi.q) + qf_frag(bh, epi.Npad, n, d0 >> 4, (d0 >> 3) & 1)) = v;
      else
        *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(epi.k) + (bh * epi.Npad + n) * 64 + d0) = v;
    }
  };
  if constexpr (AS_GEMM_EPI_PIPE != 0 && AS_GEMM_ABLATE == 0) {
    // Pipelined write-out: the tile leaves in RI slices of WM x 32 rows (the i-th 32-row block of every wave row).  Slice i is
    // converted and staged, ONE barrier, then its 16-byte stores are issued -- and while they drain the waves are already in the
    // bias / GELU / convert arithmetic of slice i + 1 (the slices use disjoint rows of the staging tile: no second barrier).
    // Before, all RI slices were converted, then all stores issued: the memory pipe idled during the first half of the
    // epilogue and the VALU during the second.
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      stage_block(i);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < WM * 32 * (BN / 8) / NT; ++t) {
        const int c = tid + t * NT;
        const int rl = c / (BN / 8), ch = c % (BN / 8);     // rl: row of the slice = (wave row, row of the 32-row block)
        store_chunk((rl >> 5) * (32 * RI) + i * 32 + (rl & 31), ch);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < (AS_GEMM_ABLATE == 8 ? 0 : RI); ++i) stage_block(i);
    __syncthreads();
    if (AS_GEMM_ABLATE == 9) return;                 // timing experiment: staging only, no copy-out
#pragma unroll
    for (int t = 0; t < BM * (BN / 8) / NT; ++t) {
      const int c = tid + t * NT;
      store_chunk(c / (BN / 8), c % (BN / 8));        // BN / 8 chunks of 8 columns per row
    }
  }
  };   // run_tile

  if constexpr (!SK) {
    const int per = (tiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= tiles) return;
    // phase offset between the two workgroups that share a CU (see launch_gemm_glds): without it both run their main loops
    // and then their epilogues at the same time; with it one's prologue / epilogue runs under the other's MFMAs
    if (epi.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
      for (int i = 0; i < epi.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    // MODE 3 (split-K): workgroup row blockIdx.y contracts over [kbeg, kbeg + klen) only and writes an fp32 partial tile
    const int kbeg = MODE == 3 ? (int)blockIdx.y * epi.N : 0;
    run_tile(tile, kbeg, MODE == 3 ? min(epi.N, K - kbeg) : K, 1, 0, 0);
  } else {
    // rank = position of this workgroup in the flattened order.  Workgroup ids are dealt round-robin over the 8 XCDs, so
    // the ranks of one XCD are made contiguous (id -> (id % 8) * (S / 8) + id / 8, S % 8 == 0): consecutive tiles -- which
    // share their A rows -- stay in one L2, as in the plain grid
    const int S = gridDim.x, nk = K / GK;
    const int rank = (blockIdx.x & 7) * (S >> 3) + (blockIdx.x >> 3);
    const long long total = (long long)tiles * nk;
    const int base = (int)(total / S), rem = (int)(total % S);
    auto start_of = [&](int r) { return (long long)r * base + (r < rem ? r : rem); };
    auto rank_of = [&](long long st) {                  // the rank whose range holds step st
      const long long cut = (long long)rem * (base + 1);
      return st < cut ? (int)(st / (base + 1)) : rem + (int)((st - cut) / base);
    };
    long long st = start_of(rank);
    const long long end = start_of(rank + 1);
    while (st < end) {
      const int tile = (int)(st / nk), k0 = (int)(st - (long long)tile * nk);
      const int n_t = (int)((end - st) < (long long)(nk - k0) ? (end - st) : (long long)(nk - k0));
      const long long t_begin = (long long)tile * nk;
      const int r_first = rank_of(t_begin), r_last = rank_of(t_begin + nk - 1);
      run_tile(tile, k0 * GK, n_t * GK, r_last - r_first + 1, rank - r_first, r_first);
      st += n_t;
      if (st < end) __syncthreads();                    // the next segment's first LDS-DMA lands in the ring / staging tile
    }
  }
}

template <int MODE, int WM, int WN, int RI = 2, int KS = 2, int NJ = 2>
int launch_gemm_glds_wm(const void* A, const void* W, const float* bias, void* out, int M, int Nout, int K, int act,
                        QkvEpi epi, hipStream_t s) {
  using GT = GTile<WM, WN, RI, KS, NJ>;
  const int tiles = as_ceil_div(M, GT::BM_) * as_ceil_div(Nout, GT::BN_);
  dim3 grid(8 * as_ceil_div(tiles, 8), MODE == 3 ? as_ceil_div(K, epi.N) : 1);   // x padded to a multiple of the 8 XCDs (tile order)
  // ring 48 / 72 / 96 KiB; the epilogue restages the output tile in the same memory (35 / 70 / 133 KiB)
  const size_t lds = (size_t)GT::LDS;
  static std::atomic<bool> attr_set{false};   // (idempotent attribute call: a race only repeats it)
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_glds_kernel<MODE, WM, WN, RI, KS, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_glds_kernel<MODE, WM, WN, RI, KS, NJ>), grid, dim3(GT::NT_), lds, s, (const __bf16*)A, (const __bf16*)W, bias,
                     (__bf16*)out, M, Nout, K, act, epi);
  AS_CHECK_LAUNCH("gemm_glds");
  return AS_OK;
}

// Tile shape, from four instantiations of one kernel:
//   short   128 x 128, 4 waves of 64 x 64, K step 32, 3-stage ring (48 KiB, 3 workgroups per CU)
//   tall    256 x 128, 8 waves of 64 x 64, K step 32, 3-stage ring (72 KiB, 2 per CU)
//   tall64  256 x 128, 8 waves of 64 x 64, K step 64, 2 stages     (96 KiB, 1 per CU)     K % 64 == 0
//   wide64  256 x 256, 8 waves of 128 x 64, K step 64, 2 stages    (136 KiB with the staged epilogue, 1 per CU)  not the QKV mode
// chosen by a per-CU round model in units of one short tile's work: a CU works through ceil(tiles / 256) tiles; a tall tile
// is two short ones done 1.28x as fast (4096^3: 662 -> 846 TFLOP/s); when the tall tiles fit ONE round (<= 256) tall64 runs
// them 4-8 % faster (whole 128-byte lines per LDS-DMA request, half the barriers; M = 8394: N = 768, K = 768: 18.4 -> 17.7 us,
// K = 3072: 55.3 -> 51.2) and the second resident workgroup it gives up has nothing to run anyway; a wide tile is four short
// ones at 1.43x (2/3 of the L2 -> LDS bytes per flop, 3/4 of the LDS reads per MFMA): it wins when it saves rounds --
// M = 8394: N = 3072, K = 768 (fc1) 396 wide tiles, 2 rounds: 59.8 -> 52.6 us; N = 1024 (ViT-L proj / fc2) 132 tiles:
// 33.6 -> 31.4, 113 -> 102; N = 4096 (ViT-L fc1) 528 tiles would need 3 rounds: tall (93.8 vs 95).  M = 8192, N = 512: 128
// tall tiles would leave half the CUs idle: short.  QKV (2304 columns) is 297 wide tiles = 2 rounds of 2.8 against 3 tall
// rounds of 1.56: tall (measured 45-47 vs 49.5 us).  Measured and not kept (tools/experiments/gemm_variant_bench.py,
// gemm_pingpong.hip.inc): the 256 x 256 tile on 16 waves of 64 x 64 (K step 32: 1000-1039 TFLOP/s at 4096^3, K step 64: 1059;
// wide64: 1106-1143), ring depths 3 / 4 / 5 for it (equal), and a ping-pong schedule of wide64 (1062-1140).
// AS_GEMM_TILE=short|tall|tall64|wide64 forces one (experiments; wide64 not in the QKV mode, *64 only if K % 64 == 0).
template <int MODE>
int launch_gemm_glds(const void* A, const void* W, const float* bias, void* out, int M, int Nout, int K, int act,
                     QkvEpi epi, hipStream_t s) {
  static const int forced = [] {
    const char* e = getenv("AS_GEMM_TILE");
    if (e == nullptr) return 0;
    return !strcmp(e, "short") ? 1 : !strcmp(e, "tall") ? 2 : !strcmp(e, "tall64") ? 3 : !strcmp(e, "wide64") ? 4 :
           !strcmp(e, "wide32") ? 5 : !strcmp(e, "tall4") ? 6 : !strcmp(e, "wide4") ? 7 : 0;
  }();
  static const int stagger = [] { const char* e = getenv("AS_GEMM_STAGGER"); return e ? atoi(e) : 0; }();
  epi.stagger = stagger;
  const int nt_n = as_ceil_div(Nout, BN), tall_tiles = as_ceil_div(M, 256) * nt_n;
  const bool k64 = K % 64 == 0;
  const float short_cost = (float)as_ceil_div(as_ceil_div(M, 128) * nt_n, 256);
  const float tall_cost = (float)as_ceil_div(tall_tiles, 256) * (tall_tiles <= 256 && k64 ? 1.47f : 2.0f / 1.28f);
  const float wide_cost = (MODE != 1 && k64) ? (float)as_ceil_div(as_ceil_div(M, 256) * as_ceil_div(Nout, 256), 256) * 2.8f : 1e30f;
  int pick = wide_cost < tall_cost && wide_cost < short_cost ? 4 : tall_cost < short_cost ? (tall_tiles <= 256 && k64 ? 3 : 2) : 1;
  if (forced && (forced < 3 || forced == 5 || forced == 6 || k64) && ((forced != 4 && forced != 5 && forced != 7) || MODE != 1)) pick = forced;
  if (pick == 6) return launch_gemm_glds_wm<MODE, 2, 2, 4, 2>(A, W, bias, out, M, Nout, K, act, epi, s);
  if constexpr (MODE != 1) {
    if (pick == 4) return launch_gemm_glds_wm<MODE, 2, 4, 4, 4>(A, W, bias, out, M, Nout, K, act, epi, s);
    // experiment (AS_GEMM_TILE=wide32): the 256 x 256 tile on K-step-32 stages, AS_GEMM_K32_STAGES deep (32 KiB each)
    if (pick == 5) return launch_gemm_glds_wm<MODE, 2, 4, 4, 2>(A, W, bias, out, M, Nout, K, act, epi, s);
    // experiment (AS_GEMM_TILE=wide4): the 256 x 256 tile on FOUR waves of 128 x 128 (one wave per SIMD, 256 accumulator registers)
    if (pick == 7) return launch_gemm_glds_wm<MODE, 2, 2, 4, 4, 4>(A, W, bias, out, M, Nout, K, act, epi, s);
  }
  if (pick == 3) return launch_gemm_glds_wm<MODE, 4, 2, 2, 4>(A, W, bias, out, M, Nout, K, act, epi, s);
  if (pick == 2) return launch_gemm_glds_wm<MODE, 4, 2, 2, 2>(A, W, bias, out, M, Nout, K, act, epi, s);
  return launch_gemm_glds_wm<MODE, 2, 2, 2, 2>(A, W, bias, out, M, Nout, K, act, epi, s);
}

// ---- stream-K launch (MODE 4 plain / 5 QKV) on the one-workgroup-per-CU tiles: S = CU count workgroups, each with the same
// number of K steps.  Worth it when the plain grid's last round is far from full: fc1 at M = 8394 is 396 tiles of 256 x 256 =
// 1.55 rounds run as 2, fc2 198 tiles of 256 x 128 on 256 CUs.  Workspace: two fp32 tile images per workgroup (a range
// starts with at most one tail piece and ends with at most one head piece) + the arrive / done counters of every tile.
struct SkPlan { int pick, tiles, S; size_t part_bytes, total; bool use; };
SkPlan sk_plan(int M, int Nout, int K, bool qkv) {
  SkPlan p{};
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  p.S = cus - cus % 8;                                   // (ranks are dealt per XCD: a multiple of 8)
  const bool wide = !qkv && Nout >= 1024 && K % 64 == 0; // the tile launch_gemm_glds would pick among the 1-per-CU ones
  p.pick = wide ? 4 : 3;
  const int bm = 256, bn = wide ? 256 : 128;
  p.tiles = as_ceil_div(M, bm) * as_ceil_div(Nout, bn);
  p.part_bytes = (size_t)p.S * 2 * bm * bn * sizeof(float);
  p.total = p.part_bytes + (size_t)as_round_up(2 * p.tiles * (int)sizeof(int), 256);
  const int rounds = as_ceil_div(p.tiles, p.S);
  // MEASURED SLOWER, so OFF unless AS_GEMM_SK=1 (round 5, one MI355X, us, plain grid -> stream-K): fc1 8394 x 3072 x 768
  // 60.1 -> 149.8, fc2 8394 x 768 x 3072 53.0 -> 94.5, proj 8394 x 768 x 768 18.6 -> 40.0.  Every workgroup publishes a
  // 128 / 256 KiB fp32 tile image behind an agent-scope release (a write-back of its XCD's whole L2, in which the other
  // workgroups' output tiles sit) and the finisher re-reads it; the second and third segment of a range restart the LDS-DMA
  // pipeline.  hipBLASLt's stream-K kernels win 25 % on these shapes with the same idea, so the cost is in THIS hand-off,
  // not in the schedule -- kept for the record and for a write-through (sc1) hand-off to be tried on it.
  const char* e = getenv("AS_GEMM_SK");                  // (read per call: tests switch it)
  const int force = e ? atoi(e) : 0;
  (void)rounds;
  p.use = force == 1 && K % 64 == 0 && p.S >= 8 && (long long)p.tiles * (K / 64) >= 8LL * p.S;
  return p;
}

template <int MODE>
int launch_gemm_sk(const void* A, const void* W, const float* bias, void* out, int M, int Nout, int K, int act, QkvEpi epi,
                   void* ws, const SkPlan& p, hipStream_t s) {
  char* w = (char*)ws;
  int* counters = (int*)(w + p.part_bytes);
  (void)hipMemsetAsync(counters, 0, (size_t)2 * p.tiles * sizeof(int), s);
  float* parts = (float*)w;
  auto go = [&](auto kern, size_t lds, int nt) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.S), dim3(nt), lds, s, (const __bf16*)A, (const __bf16*)W, bias, (__bf16*)out, M, Nout, K, act, epi);
  };
  // (the stream-K hand-off borrows the q / k pointer slots of the epilogue descriptor: MODE 4 has no QKV outputs; MODE 5
  //  carries them in epi.vt's neighbours -- see SkQkv below)
  if (p.pick == 4) {
    if constexpr (MODE == 4) {
      epi.q = parts; epi.k = counters;
      go(gemm_glds_kernel<4, 2, 4, 4, 4, 2>, (size_t)GTile<2, 4, 4, 4, 2>::LDS, GTile<2, 4, 4, 4, 2>::NT_);
    } else {
      return AS_E_UNSUPPORTED;
    }
  } else {
    if constexpr (MODE == 4) {
      epi.q = parts; epi.k = counters;
      go(gemm_glds_kernel<4, 4, 2, 2, 4, 2>, (size_t)GTile<4, 2, 2, 4, 2>::LDS, GTile<4, 2, 2, 4, 2>::NT_);
    } else {
      return AS_E_UNSUPPORTED;
    }
  }
  AS_CHECK_LAUNCH("gemm_sk");
  return AS_OK;
}

template <typename T, int MODE>
int launch_gemm(const void* A, const void* W, const float* bias, void* out, int M, int Nout, int K, int act,
                QkvEpi epi, hipStream_t s) {
  dim3 grid(as_ceil_div(M, BM), as_ceil_div(Nout, BN));
  const size_t lds = 2 * (size_t)BM * GemmCfg<T>::PITCH;
  hipLaunchKernelGGL((gemm_kernel<T, MODE>), grid, dim3(NT), lds, s, (const T*)A, (const T*)W, bias, (T*)out, M,
                     Nout, K, act, epi);
  AS_CHECK_LAUNCH("gemm");
  return AS_OK;
}

// out[e] = sum over the S partials in split order (fixed order: deterministic), converted to T
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, T* __restrict__ out, size_t n4, int S,
                                                            size_t stride) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
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

// K ranges of the split-K form: about ONE workgroup per CU (<= 256 in all, <= 32 ranges), each range a multiple of the K
// step and at least 256 long.  Fewer, fatter ranges beat filling every resident slot: the fp32 partials cross HBM twice
// (S x output x 4 bytes each way) and every workgroup pays its ring fill and a 64 KiB store -- measured on the shapes of
// a training step (tools/experiments/splitk_bench.py, incl. the reduction): 512 / 768 slots -> 256: dW_fc1 72.5 -> 61.9 us,
// dW_qkv 60.0 -> 50.5, dW_proj 45.6 -> 35.2, a head's [1024 x 256] over 51 200 rows 78.7 -> 55.4; 128 or 1024 are worse.
// Tile: 256 x 128 when the output has at least 48 of them (the backbone's weight gradients: a CU then holds 256 x 128
// instead of 128 x 128 of output at 1.28x the rate, launch_gemm_glds), else 128 x 128 (the heads' [1024 x 256] is 16 tiles).
int splitk_chunk(int M, int Nout, int K, int* splits, bool* tall = nullptr) {
  const int tall_tiles = as_ceil_div(M, 256) * as_ceil_div(Nout, 128);
  const bool use_tall = tall_tiles >= 48;
  if (tall) *tall = use_tall;
  const int tiles = use_tall ? tall_tiles : as_ceil_div(M, 128) * as_ceil_div(Nout, 128);
  static const int slots_env = [] { const char* e = getenv("AS_SPLITK_SLOTS"); return e ? atoi(e) : 0; }();   // (experiments)
  const int slots = slots_env > 0 ? slots_env : 256;
  int S = slots / (tiles > 0 ? tiles : 1);
  if (S > 32) S = 32;
  if (S < 1) S = 1;
  int chunk = as_round_up(as_ceil_div(K, S), GK);
  if (chunk < 256) chunk = 256;
  if (chunk > K) chunk = K;
  *splits = as_ceil_div(K, chunk);
  return chunk;
}

}  // namespace

// ---- csrc/gemm_pp.hip: the persistent ping-pong kernel (round 6).  cfg 0 = 256 x 256 tiles, 1 = 256 x 128 ----
bool as_pp_applies(int M, int Nout, int K, int cfg);
int as_pp_linear(const void* x, const void* W, const float* bias, void* out, void* pre, int M, int Nout, int K, int act, int cfg,
                 hipStream_t s);
int as_pp_deconv(const void* x, const void* W4, const float* bias4, void* out, int M, int w, int cin, int cout, int act, int cfg,
                 hipStream_t s);
int as_pp_qkv(const void* x, const void* Wqkv, const float* bqkv, void* q, void* k, void* vt, int B, int N, int Npad, int D, int h,
              hipStream_t s);
// which tile shape of the persistent kernel a linear takes (-1: the one-tile-per-workgroup kernels above).
// AS_GEMM_PP = 0 off | a | b force cfg 0 / 1 where it applies | unset: the round model below.  Re-read per call when
// AS_GEMM_PP_DYN is set (tools/experiments A/B in one process).
// Round model, calibrated on one MI355X (profiles/r06_gemm_pp.md): a workgroup per CU walks ceil(tiles / CUs) tiles; a K step of
// 64 costs 1.55 us on a 256 x 256 tile and 0.85 us on a 256 x 128 tile (both are bound by the L2 -> LDS stream, ~17 TB/s over
// the chip), a tile's epilogue 3 / 2 us (+ 5 / 2.5 us with the GELU).  Small problems (< 64 tiles of 256 x 128) stay with
// the one-tile-per-workgroup kernels: their launch has no pipeline to fill.
static int pp_pick(int M, int Nout, int K, bool qkv, int act) {
  static const bool dyn = getenv("AS_GEMM_PP_DYN") != nullptr;
  static const char* e0 = getenv("AS_GEMM_PP");
  const char* e = dyn ? getenv("AS_GEMM_PP") : e0;
  if (e != nullptr && e[0] == '0') return -1;
  if (e != nullptr && e[0] == 'a') return !qkv && as_pp_applies(M, Nout, K, 0) ? 0 : -1;
  if (e != nullptr && e[0] == 'b') return as_pp_applies(M, Nout, K, 1) ? 1 : -1;
  if (!as_pp_applies(M, Nout, K, 1)) return -1;
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? n - n % 8 : 8;
  }();
  const int panels = as_ceil_div(M, 256), nk = K / 64;
  const int tiles_b = panels * (Nout / 128);
  if (tiles_b < 64) return -1;
  const float gelu = act == 1 ? 1.0f : 0.0f;
  const float cost_b = (float)as_ceil_div(tiles_b, cus) * (nk * 0.85f + 2.0f + 2.5f * gelu);
  if (qkv || !as_pp_applies(M, Nout, K, 0)) return 1;
  const float cost_a = (float)as_ceil_div(panels * (Nout / 256), cus) * (nk * 1.55f + 3.0f + 5.0f * gelu);
  return cost_a < cost_b ? 0 : 1;
}

extern "C" size_t as_linear_splitk_workspace_bytes(int M, int Nout, int K) {
  if (M <= 0 || Nout <= 0 || K <= 0) return 0;
  int S = 1;
  (void)splitk_chunk(M, Nout, K, &S);
  return (size_t)S * M * Nout * sizeof(float);
}

extern "C" int as_linear_splitk_fwd(const void* x, const void* W, void* out, int M, int Nout, int K, int dtype, int out_f32,
                                    void* workspace, size_t workspace_bytes, as_stream_t stream) {
  AS_REQUIRE(x && W && out, AS_E_BADARG, "as_linear_splitk_fwd: null pointer");
  AS_REQUIRE(M > 0 && Nout > 0 && K > 0 && K % GK == 0 && Nout % 4 == 0, AS_E_BADARG,
             "as_linear_splitk_fwd: need M, N > 0, K %% 32 == 0, N %% 4 == 0 (N=%d K=%d)", Nout, K);
  AS_REQUIRE(dtype == AS_BF16, AS_E_UNSUPPORTED, "as_linear_splitk_fwd: bf16 operands only (dtype %d)", dtype);
  int S = 1;
  bool tall = false;
  const int chunk = splitk_chunk(M, Nout, K, &S, &tall);
  AS_REQUIRE(workspace && workspace_bytes >= (size_t)S * M * Nout * sizeof(float), AS_E_BADARG,
             "as_linear_splitk_fwd: workspace too small (%zu < %zu)", workspace_bytes, (size_t)S * M * Nout * sizeof(float));
  hipStream_t s = (hipStream_t)stream;
  QkvEpi epi{workspace, nullptr, nullptr, chunk, 0, 0, 0};
  static const bool no_tall = getenv("AS_SPLITK_SHORT") != nullptr;         // (experiments: always the 128 x 128 tile)
  const int rc = tall && !no_tall ? launch_gemm_glds_wm<3, 4, 2, 2, 2>(x, W, nullptr, nullptr, M, Nout, K, 0, epi, s)
                                  : launch_gemm_glds_wm<3, 2, 2, 2, 2>(x, W, nullptr, nullptr, M, Nout, K, 0, epi, s);
  if (rc != AS_OK) return rc;
  const size_t n4 = (size_t)M * Nout / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  if (out_f32)
    hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)workspace, (float*)out, n4, S,
                       (size_t)M * Nout);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const float*)workspace, (__bf16*)out, n4, S,
                       (size_t)M * Nout);
  AS_CHECK_LAUNCH("splitk_reduce");
  return AS_OK;
}

extern "C" int as_npad(int N) { return as_round_up(N, 64); }


extern "C" int as_linear_fwd(const void* x, const void* W, const float* bias, void* out, int M, int Nout, int K,
                             int dtype, int act, as_stream_t stream) {
  AS_REQUIRE(x && W && out, AS_E_BADARG, "as_linear_fwd: null pointer");
  AS_REQUIRE(M > 0 && Nout > 0 && K > 0 && K % BK == 0, AS_E_BADARG, "as_linear_fwd: need M,N>0 and K %% 32 == 0 (K=%d)", K);
  AS_REQUIRE(act == 0 || act == 1 || act == 4, AS_E_BADARG, "as_linear_fwd: act must be 0 (none), 1 (GELU) or 4 (ReLU)");
  QkvEpi epi{};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) {
    const int cfg = pp_pick(M, Nout, K, false, act);
    if (cfg >= 0) return as_pp_linear(x, W, bias, out, nullptr, M, Nout, K, act, cfg, s);
  }
  if (dtype == AS_BF16 && K % GK == 0) return launch_gemm_glds<0>(x, W, bias, out, M, Nout, K, act, epi, s);
  if (dtype == AS_BF16) return launch_gemm<__bf16, 0>(x, W, bias, out, M, Nout, K, act, epi, s);
  if (dtype == AS_F32) return launch_gemm<float, 0>(x, W, bias, out, M, Nout, K, act, epi, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_linear_fwd: dtype %d", dtype);
}

extern "C" size_t as_linear_sk_workspace_bytes(int M, int Nout, int K) {
  if (M <= 0 || Nout <= 0 || K <= 0) return 0;
  const SkPlan p = sk_plan(M, Nout, K, false);
  return p.use ? p.total : 0;
}

extern "C" int as_linear_sk_fwd(const void* x, const void* W, const float* bias, void* out, int M, int Nout, int K, int dtype,
                                int act, void* ws, size_t ws_bytes, as_stream_t stream) {
  AS_REQUIRE(x && W && out, AS_E_BADARG, "as_linear_sk_fwd: null pointer");
  AS_REQUIRE(M > 0 && Nout > 0 && K > 0 && K % BK == 0, AS_E_BADARG, "as_linear_sk_fwd: need M,N>0 and K %% 32 == 0 (K=%d)", K);
  AS_REQUIRE(act == 0 || act == 1 || act == 4, AS_E_BADARG, "as_linear_sk_fwd: act must be 0 (none), 1 (GELU) or 4 (ReLU)");
  const SkPlan p = sk_plan(M, Nout, K, false);
  if (dtype != AS_BF16 || !p.use) return as_linear_fwd(x, W, bias, out, M, Nout, K, dtype, act, stream);
  AS_REQUIRE(ws && ws_bytes >= p.total, AS_E_WORKSPACE, "as_linear_sk_fwd: workspace %zu < %zu bytes", ws_bytes, p.total);
  return launch_gemm_sk<4>(x, W, bias, out, M, Nout, K, act, QkvEpi{}, ws, p, (hipStream_t)stream);
}

extern "C" int as_linear_gelu_fwd(const void* x, const void* W, const float* bias, void* out, void* pre, int M, int Nout, int K,
                                  int dtype, as_stream_t stream) {
  AS_REQUIRE(x && W && out && pre, AS_E_BADARG, "as_linear_gelu_fwd: null pointer");
  AS_REQUIRE(M > 0 && Nout > 0 && K > 0 && K % GK == 0 && Nout % 8 == 0, AS_E_BADARG,
             "as_linear_gelu_fwd: need M > 0, K %% 32 == 0, Nout %% 8 == 0 (Nout=%d K=%d)", Nout, K);
  AS_REQUIRE(dtype == AS_BF16, AS_E_UNSUPPORTED, "as_linear_gelu_fwd: bf16 only (dtype %d)", dtype);
  {
    const int cfg = pp_pick(M, Nout, K, false, 1);
    if (cfg >= 0) return as_pp_linear(x, W, bias, out, pre, M, Nout, K, 2, cfg, (hipStream_t)stream);
  }
  QkvEpi epi{pre, nullptr, nullptr, 0, 0, 0, 0};
  return launch_gemm_glds<0>(x, W, bias, out, M, Nout, K, 2, epi, (hipStream_t)stream);
}

extern "C" int as_linear_dgelu_fwd(const void* x, const void* W, const void* pre, void* out, int M, int Nout, int K, int dtype,
                                   as_stream_t stream) {
  AS_REQUIRE(x && W && out && pre, AS_E_BADARG, "as_linear_dgelu_fwd: null pointer");
  AS_REQUIRE(M > 0 && Nout > 0 && K > 0 && K % GK == 0 && Nout % 8 == 0, AS_E_BADARG,
             "as_linear_dgelu_fwd: need M > 0, K %% 32 == 0, Nout %% 8 == 0 (Nout=%d K=%d)", Nout, K);
  AS_REQUIRE(dtype == AS_BF16, AS_E_UNSUPPORTED, "as_linear_dgelu_fwd: bf16 only (dtype %d)", dtype);
  {
    const int cfg = pp_pick(M, Nout, K, false, 1);
    if (cfg >= 0) return as_pp_linear(x, W, nullptr, out, const_cast<void*>(pre), M, Nout, K, 3, cfg, (hipStream_t)stream);
  }
  QkvEpi epi{const_cast<void*>(pre), nullptr, nullptr, 0, 0, 0, 0};
  return launch_gemm_glds<0>(x, W, nullptr, out, M, Nout, K, 3, epi, (hipStream_t)stream);
}

extern "C" int as_deconv2x2_fwd(const void* x, const void* W4, const float* bias4, void* out, int M, int w, int cin, int cout,
                                int dtype, int act, as_stream_t stream) {
  AS_REQUIRE(x && W4 && out, AS_E_BADARG, "as_deconv2x2_fwd: null pointer");
  AS_REQUIRE(M > 0 && w > 0 && M % w == 0 && cin > 0 && cout > 0, AS_E_BADARG, "as_deconv2x2_fwd: bad sizes");
  AS_REQUIRE(dtype == AS_BF16 && cin % GK == 0 && cout % 8 == 0, AS_E_UNSUPPORTED,
             "as_deconv2x2_fwd: bf16 with cin %% 32 == 0 and cout %% 8 == 0 only (cin=%d cout=%d dtype=%d)", cin, cout, dtype);
  AS_REQUIRE(act == 0 || act == 1, AS_E_BADARG, "as_deconv2x2_fwd: act must be 0 or 1");
  {
    const int cfg = pp_pick(M, 4 * cout, cin, false, act);
    if (cfg >= 0 && (4 * cout) % (cfg == 0 ? 256 : 128) == 0 && cout % 8 == 0)
      return as_pp_deconv(x, W4, bias4, out, M, w, cin, cout, act, cfg, (hipStream_t)stream);
  }
  QkvEpi epi{nullptr, nullptr, nullptr, w, 0, cout, 0};
  return launch_gemm_glds<2>(x, W4, bias4, out, M, 4 * cout, cin, act, epi, (hipStream_t)stream);
}

extern "C" int as_qkv_fwd(const void* x, const void* Wqkv, const float* bqkv, void* q, void* k, void* vt, int B,
                          int N, int D, int h, int dtype, as_stream_t stream) {
  AS_REQUIRE(x && Wqkv && q && k && vt, AS_E_BADARG, "as_qkv_fwd: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && h > 0 && D == h * AS_HEAD_DIM, AS_E_UNSUPPORTED,
             "as_qkv_fwd: head dim must be 64 (D=%d h=%d)", D, h);
  QkvEpi epi{q, k, vt, N, as_npad(N), D, h};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16 && D % 128 == 0 && pp_pick(B * N, 3 * D, D, true, 0) == 1)
    return as_pp_qkv(x, Wqkv, bqkv, q, k, vt, B, N, as_npad(N), D, h, s);
  if (dtype == AS_BF16 && D % GK == 0) return launch_gemm_glds<1>(x, Wqkv, bqkv, nullptr, B * N, 3 * D, D, 0, epi, s);
  if (dtype == AS_BF16) return launch_gemm<__bf16, 1>(x, Wqkv, bqkv, nullptr, B * N, 3 * D, D, 0, epi, s);
  if (dtype == AS_F32) return launch_gemm<float, 1>(x, Wqkv, bqkv, nullptr, B * N, 3 * D, D, 0, epi, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_qkv_fwd: dtype %d", dtype);
}
