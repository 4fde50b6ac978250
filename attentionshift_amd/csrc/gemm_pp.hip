// Persistent ping-pong GEMM for gfx950 (round 6):  C[M,Nout] = act(A[M,K] . W[Nout,K]^T + bias), bf16 operands.
//
// Replaces nn.Linear / the QKV projection of Attention and Mlp (reference models/vision_transformer.py:47-59, 75-77, 84)
// on the shapes of the backbone (M = B * N tokens, K and Nout multiples of 64 / 128).  What is different from the
// one-tile-per-workgroup kernel of gemm.hip, and why (profiles/r04_gemm_analysis.md: 39 % of an fc1 launch was prologue +
// epilogue, the main loop 71 % busy):
//   * ONE workgroup per CU that walks a list of output tiles, and ONE operand stream over all of its (tile, K step) pairs:
//     the LDS-DMA prefetch of the next tile's first K steps is issued under the previous tile's last K steps, so a tile has
//     no prologue, and its global stores drain under the next tile's main loop (they are fire-and-forget);
//   * the epilogue goes from the accumulators straight to memory: the W rows of a tile are PERMUTED when they are staged
//     (the LDS-DMA source address is per lane, so this is free) such that the two 16 x 16 MFMA fragments of a 32-column block
//     leave every lane with 8 CONSECUTIVE output columns = one 16-byte store; no LDS staging tile, no barrier;
//   * the main loop is the two-group ("ping-pong") schedule on v_mfma_f32_16x16x32_bf16: waves 0-3 and 4-7 (one of each
//     per SIMD) run half a phase apart, so that one group's fragment reads + LDS-DMA issue sit under the other group's
//     MFMAs; every phase is {ds_read the phase's fragments, issue one half-tile of LDS-DMA, s_barrier, lgkmcnt(0), 16 MFMAs,
//     s_barrier}; the DMA is waited for with a counted vmcnt ONCE per K step, never 0 (cdna_hip_programming.md section 5,
//     "8-phase" schedule; the phase tables below are this kernel's own and are derived in DESIGN.md section 4.2).
// Two tile shapes:
//   cfg 0  256 x 256 x 64, waves 2 (M) x 4 (N), wave tile (64 + 64) x (32 + 32), 4 phases per K step, 2 LDS buffers x 64 KiB
//   cfg 1  256 x 128 x 64, waves 4 (M) x 2 (N), wave tile 64 x (32 + 32),        2 phases per K step, 3 LDS buffers x 48 KiB
// LDS image of a half-tile (16 KiB = 128 rows x 64 k): row r at r * 128, 16-byte chunk c of the row at slot c ^ ((r >> 1) & 7):
// the 16-lane groups of a ds_read_b128 fragment read (rows i = lane & 15, chunk lane >> 4) then cover the 16 slots of a 256-byte
// bank row exactly once.  The image is lane-linear for the LDS-DMA, so the XOR sits on the SOURCE chunk.
#include <map>
#include <mutex>
#include <utility>
#include "common.h"
#include "gemm_epi.h"

#ifndef AS_PP_PRIO
#define AS_PP_PRIO 1     // s_setprio: 0 none, 1 raised around the MFMA cluster, 2 raised in the load segment
#endif
#ifndef AS_PP_BUF
#define AS_PP_BUF 0      // LDS-DMA instruction: 0 global_load_lds (a 64-bit address per lane), 1 buffer_load ... lds (one descriptor per
#endif                   // operand, a 32-bit offset per lane that is fixed per tile, the K step in the scalar offset).  Measured equal
                         // (the stream runs at ~17 TB/s over the chip either way, profiles/r06_gemm_pp.md); 0 is 0-5 % ahead on 256 x 256
// (the buffer-descriptor type exists in the device pass only: the host pass, which needs the kernel for its launch stub only,
//  compiles the pointer form)
#if AS_PP_BUF && defined(__HIP_DEVICE_COMPILE__)
#define PP_BUF 1
#else
#define PP_BUF 0
#endif
#ifndef AS_PP_ABLATE
#define AS_PP_ABLATE 0   // timing experiments (tools/experiments/gemm_pp_ablate.py); results are WRONG when != 0.  Bit mask:
#endif                   // 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no MFMAs, 8 no bank swizzle on the DMA source

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned pp_u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 pp_bf16x2;

struct PPEpi {
  void* q; void* k; void* vt;       // QKV mode outputs
  int N, Npad, D, h;                // tokens per image, padded, model width, heads
};

// Stream-K plan of a launch (SK kernels only; csrc/gemm_pp.hip launch_pp): the first D * G tiles are whole tiles (workgroup r:
// r, r + G, ...), the other `sk_tiles` (G <= sk_tiles < 2 G) are cut into G contiguous ranges of the (tile, K step) sequence.
struct PPPlan {
  int D, sk_tiles;
  unsigned epoch;                   // this launch's flag value (never 0; flags are never reset)
  void* slabs;                      // [G][NV4][512] 16-byte vectors: the fp32 accumulators of a workgroup's open tile
  unsigned* flags;                  // [G][8]: wave w of workgroup g has published its part of slab g when flags[g][w] == epoch
};

template <int OFF> __device__ __forceinline__ void pp_read(pp_u32x4& d, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
#if AS_PP_ABLATE & 2
  asm volatile("" : "=v"(d) : "v"(addr));
#else
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
#endif
}
__device__ __forceinline__ void pp_wait12(pp_u32x4 (&x)[4][2], pp_u32x4 (&w)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]),
                 "+v"(x[3][1]), "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]));
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_wait8(pp_u32x4 (&x)[4][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]),
                 "+v"(x[3][1]));
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pp_wait4(pp_u32x4 (&w)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]));
  __builtin_amdgcn_sched_barrier(0);
}

// column of a 32-column block held by MFMA row j of the block's two 16-row fragments (j = 16 f + i, lane group g = i >> 2,
// register r = i & 3): 8 g + 4 f + r -- fragment pair (f = 0, 1) gives lane group g the columns 8 g .. 8 g + 7
__device__ __forceinline__ int pp_pi32(int j) { return 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3); }

template <int... I, typename F> __device__ __forceinline__ void pp_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <typename F> __device__ __forceinline__ void g_static_for16(F&& f) {
  pp_static_for_impl(std::make_integer_sequence<int, 16>{}, f);
}

template <int CFG> struct PPCfg;
template <> struct PPCfg<0> {
  static constexpr int BM = 256, BN = 256, NBUF = 2, BUF = 65536, MA = 8, NWH = 2;
  static constexpr int X0 = 0, X1 = 16384, W0 = 32768, W1 = 49152;
};
template <> struct PPCfg<1> {
  static constexpr int BM = 256, BN = 128, NBUF = 3, BUF = 49152, MA = 4, NWH = 1;
  static constexpr int X0 = 0, X1 = 16384, W0 = 32768, W1 = 32768;
};

// EM: 0 = row-major out + bias, ACT 0 none / 1 GELU / 4 ReLU / 2 training fc1 (the bf16 pre-activation goes to epi.q, its GELU to
//         out) / 3 fc2's input gradient (out = acc * GELU'(pre), pre read from epi.q);
//     1 = QKV scatter (q fragment-major pre-scaled, k, V^T);
//     2 = 2 x 2 / stride-2 transposed convolution: GEMM row = input pixel of a grid epi.N wide, column = (tap, co < epi.D) ->
//         NHWC output pixel (as_deconv2x2_fwd), ACT 0 / 1
// (Where the LDS-DMA instructions are issued was measured three ways, profiles/r06_gemm_pp.md: in the phase's load segment
//  beside the fragment reads -- kept --, between the MFMAs of the phase's cluster, and behind the cluster; the last two are
//  12-17 % slower on every shape: the issuing wave's MFMAs stall behind each DMA instruction.)
// SK (stream-K tail, round 6): the tiles that do not fill a whole round of the G workgroups are shared out by K steps, so that
// every workgroup streams the same number of K steps (+- 1).  A workgroup's range [u0, u1) of the tail's (tile, K step)
// sequence is 1 .. 2 tiles long (G <= sk_tiles < 2 G), so it touches 2 or 3 tiles and a tile has at most TWO contributors:
//   * the workgroup that holds a tile's FIRST K steps but not its last runs that piece FIRST, stores the accumulators to its
//     slab (write-through 16-byte stores) and publishes per-wave flags two K steps later -- loads and stores retire in order
//     on the vm counter, so the K loop's own counted wait covers the slab stores by then: no drain, no fence;
//   * then its whole tiles; and LAST the tile whose remaining K steps it holds: it starts from the neighbour's slab instead of
//     zero (flag poll + sc1 loads; the slab was written ~a whole launch earlier) and finishes the tile.
// Every output is still ONE k-ordered fp32 MFMA chain -- handed over once through memory -- so the result is bit-identical to the
// non-split kernel's and to gemm.hip's.  No workgroup ever waits for a workgroup that waits: producers publish before anything else.
template <int CFG, int EM, int ACT, bool SK>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W,
                                                         const float* __restrict__ bias, __bf16* __restrict__ out, int M,
                                                         int Nout, int K, PPEpi epi, PPPlan plan) {
#if defined(__HIP_DEVICE_COMPILE__)                            // (buffer descriptors are a device-pass type; the host pass needs the stub only)
  using C = PPCfg<CFG>;
  constexpr int BM = C::BM, BN = C::BN, NBUF = C::NBUF, MA = C::MA;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                 // ping-pong group: waves w and w + 4 share a SIMD
  const int wr = CFG == 0 ? (wave >> 2) : (wave >> 1);       // wave row    (cfg 0: 2, cfg 1: 4)
  const int wc = CFG == 0 ? (wave & 3) : (wave & 1);         // wave column (cfg 0: 4, cfg 1: 2)
  const int li = lane & 15, lg = lane >> 4;

  // ---- this workgroup's tiles: rank r of G (ranks of one XCD contiguous), tiles r, r + G, ... in (panel, column) order ----
  const int nt_n = Nout / BN, tiles = ((M + BM - 1) / BM) * nt_n;
  const int G = gridDim.x;
  const int rank = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (rank >= tiles) return;
  const int nk = K >> 6;
  // this workgroup's SEGMENTS in execution order: (tile, first K step, end K step).  Not SK: its whole tiles.  SK: [the open
  // head of a tail tile] [D whole tiles] [up to three pieces of the tail, the one that continues a neighbour's tile last]
  int nA = 0, nB = 0, nD = (tiles - rank + G - 1) / G;
  int a_t = 0, a_k1 = 0, b0_t = 0, b0_k0 = 0, b1_t = 0, b1_k0 = 0, b2_t = 0, b2_k0 = 0;   // (scalars, no indexed array: SGPRs)
  int total = nD * nk;                                       // K steps of this workgroup's operand stream
  if constexpr (SK) {
    nD = plan.D;
    const int base = plan.D * G;
    const unsigned U = (unsigned)(plan.sk_tiles * nk);        // (U * G < 2^31: checked by the launcher; 32-bit scalar divisions)
    const int u0 = __builtin_amdgcn_readfirstlane((int)(U * (unsigned)rank / (unsigned)G));
    const int u1 = __builtin_amdgcn_readfirstlane((int)(U * (unsigned)(rank + 1) / (unsigned)G));
    const int tf = u0 / nk, kf = u0 - tf * nk, tl = (u1 - 1) / nk, kl = u1 - tl * nk;
    if (tf == tl) {                                          // (one whole tile: u1 - u0 >= nk)
      b0_t = base + tf; b0_k0 = 0; nB = 1;
    } else {
      const bool tail_full = kl == nk, mid = tl - tf == 2;
      if (!tail_full) { a_t = base + tl; a_k1 = kl; nA = 1; }
      nB = 1 + (tail_full ? 1 : 0) + (mid ? 1 : 0);           // [whole tail] [middle] head
      b0_t = base + (tail_full ? tl : mid ? tf + 1 : tf);
      b0_k0 = tail_full || mid ? 0 : kf;
      b1_t = base + (tail_full && mid ? tf + 1 : tf);
      b1_k0 = tail_full && mid ? 0 : kf;
      b2_t = base + tf;
      b2_k0 = kf;
    }
    total = nD * nk + (u1 - u0);
  }
  const int nseg = nA + nD + nB;
  // segment i -> tile, k0, k1 (all uniform)
  struct Seg { int tile, k0, k1; };
  auto seg_at = [&](int i) __attribute__((always_inline)) -> Seg {
    if (SK && i < nA) return Seg{a_t, 0, a_k1};
    if (!SK || i < nA + nD) return Seg{rank + (i - nA) * G, 0, nk};
    const int j = i - nA - nD;
    // (arithmetic selects on VALUES: `c ? x : y` on captured variables is a select of their addresses -- hipcc then keeps every
    //  captured scalar in scratch memory and loads it back through a flat pointer inside the K loop)
    const int t0 = b0_t + 0, t1 = b1_t + 0, t2 = b2_t + 0, q0 = b0_k0 + 0, q1 = b1_k0 + 0, q2 = b2_k0 + 0;
    const int m0 = -(int)(j == 0), m1 = -(int)(j == 1), m2 = -(int)(j >= 2);
    return Seg{(t0 & m0) | (t1 & m1) | (t2 & m2), (q0 & m0) | (q1 & m1) | (q2 & m2), nk};
  };
  const Seg seg0 = seg_at(0);
  const int seg0_t = seg0.tile, seg0_k0 = seg0.k0, seg0_k1 = seg0.k1;

  // ---- LDS-DMA source pointers: [half-tile type][piece]; wave w moves pieces 2w, 2w+1 (8 LDS rows x 128 B each) ----
  // piece p, lane l -> LDS row 8 p + (l >> 3), physical chunk l & 7 -> source chunk (l & 7) ^ ((row >> 1) & 7)
#if PP_BUF
  unsigned sx[2][2], sw[C::NWH][2];                          // per-lane byte offsets into A / W (fixed per tile)
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(A), (short)0, (int)((size_t)M * K * 2), 0x00027000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W), (short)0, (int)((size_t)Nout * K * 2), 0x00027000);
#else
  const char* sx[2][2];                                      // X0 / X1
  const char* sw[C::NWH][2];                                 // W0 (/ W1)
#endif
  int ld_row[2], ld_chunk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    ld_row[j] = (2 * wave + j) * 8 + (lane >> 3);
    ld_chunk[j] = ((lane & 7) ^ ((AS_PP_ABLATE & 8) ? 0 : ((ld_row[j] >> 1) & 7))) << 4;
  }
  auto x_src = [&](int h, int j, int tile) __attribute__((always_inline)) {
    const int m0 = (tile / nt_n) * BM, l = ld_row[j];
    const int trow = CFG == 0 ? ((l >> 6) * 128 + h * 64 + (l & 63)) : (h * 128 + l);
#if PP_BUF
    return (unsigned)(min(m0 + trow, M - 1) * (K * 2) + ld_chunk[j]);
#else
    return reinterpret_cast<const char*>(A) + (size_t)min(m0 + trow, M - 1) * K * 2 + ld_chunk[j];
#endif
  };
  auto w_src = [&](int h, int j, int tile) __attribute__((always_inline)) {
    const int n0 = (tile % nt_n) * BN, l = ld_row[j];
    const int tcol = CFG == 0 ? ((l >> 5) * 64 + h * 32 + pp_pi32(l & 31)) : ((l >> 5) * 32 + pp_pi32(l & 31));
#if PP_BUF
    return (unsigned)((n0 + tcol) * (K * 2) + ld_chunk[j]);
#else
    return reinterpret_cast<const char*>(W) + (size_t)(n0 + tcol) * K * 2 + ld_chunk[j];
#endif
  };
  // stream position of every half-tile type: (tile index in my list, K step, LDS buffer)
  int st_ti[2 + C::NWH], st_kt[2 + C::NWH], st_ke[2 + C::NWH], st_buf[2 + C::NWH];   // (segment index, K step, the segment's end K step)
#pragma unroll
  for (int t = 0; t < 2 + C::NWH; ++t) { st_ti[t] = 0; st_kt[t] = seg0_k0; st_ke[t] = seg0_k1; st_buf[t] = 0; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sx[0][j] = x_src(0, j, seg0_t); sx[1][j] = x_src(1, j, seg0_t);
#pragma unroll
    for (int h = 0; h < C::NWH; ++h) sw[h][j] = w_src(h, j, seg0_t);
  }
  static_assert(!(SK && PP_BUF), "the stream-K kernels use the pointer form of the LDS-DMA");
  bool in_loop = false;                                      // (AS_PP_ABLATE & 1 only)
  // stage<T>(): the next K step of half-tile type T (0 X0, 1 X1, 2 W0, 3 W1) -> its slot of buffer st_buf[T], two LDS-DMA
  // instructions per wave.  The stream never runs dry: behind the last K step of the last tile it wraps to that tile's first
  // K step again (two or three K steps of loads nobody reads, into slots that are free by then) -- so every counted vmcnt below
  // holds for every iteration and the load segment carries no "stream exhausted" branches (each s_cbranch in it is on the
  // critical path of the ping-pong: the load segment, not the 16 MFMAs, bounds a phase).
  auto stage = [&](auto t_c) __attribute__((always_inline)) {
    constexpr int T = decltype(t_c)::value;
    constexpr int SLOT = T == 0 ? C::X0 : T == 1 ? C::X1 : T == 2 ? C::W0 : C::W1;
    if ((AS_PP_ABLATE & 1) && in_loop) return;
    char* dst = smem + st_buf[T] * C::BUF + SLOT + (2 * wave) * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#if PP_BUF
      unsigned off;
      if constexpr (T < 2) off = sx[T][j];
      else off = sw[T - 2][j];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(T < 2 ? rsrcA : rsrcW, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16,
                                               (int)off, st_kt[T] * 128, 0, 0);
#else
      const char* src;
      if constexpr (T < 2) src = sx[T][j];
      else src = sw[T - 2][j];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
      if constexpr (T < 2) sx[T][j] = src + 128;
      else sw[T - 2][j] = src + 128;
#endif
    }
    st_buf[T] = st_buf[T] + 1 == NBUF ? 0 : st_buf[T] + 1;
    if (__builtin_expect(++st_kt[T] == st_ke[T], 0)) {       // (once per segment and type)
      st_ti[T] = st_ti[T] + 1 < nseg ? st_ti[T] + 1 : st_ti[T];
      const Seg sg = seg_at(st_ti[T]);
      const int tile = sg.tile, k0 = sg.k0;
      st_kt[T] = k0;
      st_ke[T] = sg.k1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#if PP_BUF
        if constexpr (T < 2) sx[T][j] = x_src(T, j, tile);
        else sw[T - 2][j] = w_src(T - 2, j, tile);
#else
        if constexpr (T < 2) sx[T][j] = x_src(T, j, tile) + k0 * 128;
        else sw[T - 2][j] = w_src(T - 2, j, tile) + k0 * 128;
#endif
      }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  // ---- fragment read addresses (per lane): row i = lane & 15 of a 16-row fragment, chunk (lane >> 4) + 4 ks ----
  const unsigned smem_base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  const unsigned frag_lane = (unsigned)(li * 128 + ((lg ^ ((li >> 1) & 7)) << 4));
  const unsigned ks1 = (frag_lane & 64u) ? (unsigned)-64 : 64u;   // k32 step 1 = chunk ^ 4 = +- 64 bytes (no alignment assumed)
  const unsigned x_lane = smem_base + frag_lane + (CFG == 0 ? wr * 64 * 128 : (wr & 1) * 64 * 128 + (wr >> 1) * C::X1);
  const unsigned w_lane = smem_base + frag_lane + (CFG == 0 ? wc * 32 * 128 : wc * 64 * 128);

  f32x4 acc[MA][2][2];                                       // [m fragment][32-column block][fragment of the block]
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[a][p][f] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };
  zero_acc();

  pp_u32x4 xf[4][2], wf[2][2];                               // [fragment][ks]
  // 16 MFMAs: m fragments A0 .. A0+3 x the two fragments of column block P, two k32 steps; D[n][m] = W . X^T
  auto mfma16 = [&](auto a0_c, auto p_c) {
    constexpr int A0 = decltype(a0_c)::value, P = decltype(p_c)::value;
    if (AS_PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if (AS_PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    g_static_for16([&](auto n_c) {
      constexpr int n = decltype(n_c)::value, ks = n >> 3, a = (n >> 1) & 3, f = n & 1;
#if AS_PP_ABLATE & 4
      auto& a_ = acc[A0 + a][P][f];
      auto& w_ = wf[f][ks];
      auto& x_ = xf[a][ks];
      asm volatile("" : "+v"(a_) : "v"(w_), "v"(x_));
#else
      acc[A0 + a][P][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[f][ks]),
                                                                  __builtin_bit_cast(bf16x8, xf[a][ks]), acc[A0 + a][P][f], 0, 0, 0);
#endif
    });
    if (AS_PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
  };
  auto read_x = [&](auto slot_c, unsigned base) {             // 4 fragments x 2 k32 steps of one X half-tile
    constexpr int SLOT = decltype(slot_c)::value;
    const unsigned b1 = base + ks1;
    pp_read<SLOT + 0 * 2048>(xf[0][0], base); pp_read<SLOT + 0 * 2048>(xf[0][1], b1);
    pp_read<SLOT + 1 * 2048>(xf[1][0], base); pp_read<SLOT + 1 * 2048>(xf[1][1], b1);
    pp_read<SLOT + 2 * 2048>(xf[2][0], base); pp_read<SLOT + 2 * 2048>(xf[2][1], b1);
    pp_read<SLOT + 3 * 2048>(xf[3][0], base); pp_read<SLOT + 3 * 2048>(xf[3][1], b1);
  };
  auto read_w = [&](auto slot_c, unsigned base) {             // 2 fragments x 2 k32 steps of one 32-column block
    constexpr int SLOT = decltype(slot_c)::value;
    const unsigned b1 = base + ks1;
    pp_read<SLOT + 0 * 2048>(wf[0][0], base); pp_read<SLOT + 0 * 2048>(wf[0][1], b1);
    pp_read<SLOT + 1 * 2048>(wf[1][0], base); pp_read<SLOT + 1 * 2048>(wf[1][1], b1);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // barrier that ends an MFMA segment = opens a load segment (AS_PP_PRIO 2: the loader, not the MFMA wave, gets the issue slots)
  auto bar_load = [&]() {
    bar();
    if (AS_PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
  };

  // ---- epilogue: accumulators -> memory, 16 bytes (8 consecutive columns) per lane and (m fragment, column block) ----
  auto epilogue = [&](int tile) __attribute__((always_inline)) {
    const int m0 = (tile / nt_n) * BM, n0 = (tile % nt_n) * BN;
    const int colw = n0 + wc * 64 + 8 * lg;                  // + 32 p
    f32x4 bv[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (bias != nullptr) {
        bv[p][0] = *reinterpret_cast<const f32x4*>(bias + colw + 32 * p);
        bv[p][1] = *reinterpret_cast<const f32x4*>(bias + colw + 32 * p + 4);
      } else {
        bv[p][0] = bv[p][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    // the bias is waited for HERE, on every path: left to the first use, the wait lands inside the row < M branches -- hipcc then
    // repeats a vmcnt(0) in every branch (each store waits for the one before it) and, through the loop's back edge, puts one in
    // front of the main loop's fragment reads (the LDS-DMA queue drained every K step)
#pragma unroll
    for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(bv[p][0]), "+v"(bv[p][1]));
    int which = 0, head0 = 0;
    if constexpr (EM == 1) { which = n0 / epi.D; head0 = (n0 % epi.D) >> 6; }
    const int row0 = m0 + (CFG == 0 ? wr * 128 : wr * 64) + li;
#pragma unroll
    for (int a0 = 0; a0 < MA; a0 += 4) {
      // ACT 3: the pre-activations of these four row fragments, loaded (rows clamped: no branch) and waited for up front
      pp_u32x4 pre[4][2];
      if constexpr (EM == 0 && ACT == 3) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            pre[a][p] = *reinterpret_cast<const pp_u32x4*>(reinterpret_cast<const __bf16*>(epi.q) +
                                                           (size_t)min(row0 + 16 * (a0 + a), M - 1) * Nout + colw + 32 * p);
#pragma unroll
        for (int a = 0; a < 4; ++a) asm volatile("" : "+v"(pre[a][0]), "+v"(pre[a][1]));
      }
#pragma unroll
      for (int a = a0; a < a0 + 4; ++a) {
        const int row = row0 + 16 * a;
        int b_img = 0, n_img = row;
        if constexpr (EM == 1) { b_img = row / epi.N; n_img = row - b_img * epi.N; }
        if constexpr (EM == 2) { b_img = row / epi.N; n_img = row - b_img * epi.N; }   // (pixel row of the grid, column)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = acc[a][p][0][r] + bv[p][0][r]; v[4 + r] = acc[a][p][1][r] + bv[p][1][r]; }
          if constexpr (EM != 1) {
            if constexpr (ACT == 1) {
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const g_f32x2 gq = gelu_bf16_x2(g_f32x2{v[e], v[e + 1]});
                v[e] = gq.x; v[e + 1] = gq.y;
              }
            } else if constexpr (ACT == 4) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
            }
          } else if (which == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= AS_QSCALE;   // q is stored pre-scaled by log2(e) / 8 (common.h)
          }
          uint4 pk;
          pk.x = __builtin_bit_cast(unsigned, pp_bf16x2{(__bf16)v[0], (__bf16)v[1]});
          pk.y = __builtin_bit_cast(unsigned, pp_bf16x2{(__bf16)v[2], (__bf16)v[3]});
          pk.z = __builtin_bit_cast(unsigned, pp_bf16x2{(__bf16)v[4], (__bf16)v[5]});
          pk.w = __builtin_bit_cast(unsigned, pp_bf16x2{(__bf16)v[6], (__bf16)v[7]});
          if (row >= M) continue;
          const int col = colw + 32 * p;
          if constexpr (EM == 0) {
            if constexpr (ACT == 2) {                          // (gelu_chunk: the GELU of the ROUNDED pre-activation, as gemm.hip)
              *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(epi.q) + (size_t)row * Nout + col) = pk;
              *reinterpret_cast<uint4*>(out + (size_t)row * Nout + col) = gelu_chunk(pk);
            } else if constexpr (ACT == 3) {
              const pp_u32x4 h = pre[a - a0][p];
              *reinterpret_cast<uint4*>(out + (size_t)row * Nout + col) = dgelu_chunk(pk, make_uint4(h.x, h.y, h.z, h.w));
            } else {
              *reinterpret_cast<uint4*>(out + (size_t)row * Nout + col) = pk;
            }
          } else if constexpr (EM == 2) {
            const int tap = col / epi.D, co = col - tap * epi.D;   // (a 16-byte chunk never straddles a tap: epi.D % 8 == 0)
            *reinterpret_cast<uint4*>(out + ((size_t)(2 * b_img + (tap >> 1)) * (2 * epi.N) + 2 * n_img + (tap & 1)) * epi.D + co) = pk;
          } else {
            const int cl = col - n0;                          // column inside the 128-wide tile: head cl >> 6, d0 = cl & 63
            const size_t bh = (size_t)(b_img * epi.h + head0 + (cl >> 6));
            const int d0 = cl & 63;
            if (which == 0) {
              *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(epi.q) + qf_frag(bh, epi.Npad, n_img, d0 >> 4, (d0 >> 3) & 1)) = pk;
            } else if (which == 1) {
              *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(epi.k) + (bh * epi.Npad + n_img) * 64 + d0) = pk;
            } else {
              // V^T [B,h,64,Npad]: the lane's 8 values are 8 features of ONE token: 2-byte stores, 16 consecutive tokens
              // (32 bytes) per lane group and instruction
              __bf16* dst = reinterpret_cast<__bf16*>(epi.vt) + (bh * 64 + d0) * epi.Npad + n_img;
              const unsigned wds[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const unsigned short bits = (unsigned short)(e & 1 ? wds[e >> 1] >> 16 : wds[e >> 1] & 0xffffu);
                *reinterpret_cast<unsigned short*>(dst + (size_t)e * epi.Npad) = bits;
              }
            }
          }
        }
      }
    }
  };

  // ---- prologue: fill the pipeline (see the schedule tables in DESIGN.md section 4.2) ----
  if constexpr (CFG == 0) {
    stage(I0{}); stage(I2{}); stage(I3{}); stage(I1{});       // K step 0: X0 W0 W1 X1
    stage(I0{}); stage(I3{});                                 // K step 1: X0 W1   (X1, W0 of step 1 follow in phases 0, 1)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    stage(I0{}); stage(I1{}); stage(I2{});                    // K step 0: X0 X1 W
    stage(I0{}); stage(I1{}); stage(I2{});                    // K step 1
    stage(I0{});                                              // K step 2: X0
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  bar();
  if (grp == 1) bar();                                        // group 1 runs one barrier behind group 0
  if (AS_PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);

  // ---- stream-K hand-over of an open tile (SK): the accumulators as 16-byte vectors, vector j of thread t at [slab][j][t] ----
  constexpr int NV4 = MA * 4;
  const __amdgpu_buffer_rsrc_t rs_slab =
      __builtin_amdgcn_make_buffer_rsrc(plan.slabs, (short)0, SK ? (int)((size_t)G * NV4 * 512 * 16) : 0, 0x00027000);
  int pub = 0;                                               // > 0: K steps until this wave may publish its slab's flag
  auto slab_store = [&]() __attribute__((always_inline)) {
    const unsigned base = (unsigned)((rank * NV4) * 512 + tid) * 16u;
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int f = 0; f < 2; ++f)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pp_u32x4, acc[a][p][f]), rs_slab,
                                                 (int)(base + (unsigned)((a * 2 + p) * 2 + f) * 8192u), 0, 16 /* sc1: write-through */);
  };
  auto slab_publish = [&]() __attribute__((always_inline)) {                                // (every lane stores the same word: no divergent branch in the loop)
    __hip_atomic_store(plan.flags + rank * 8 + wave, plan.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto slab_load = [&]() __attribute__((always_inline)) {                                   // the neighbour's open tile -> the accumulators
    const unsigned* flag = plan.flags + (rank - 1) * 8 + wave;
    // (readfirstlane: a UNIFORM exit -- hipcc otherwise treats everything the K loop carries past this loop as divergent and
    //  moves the stream's counters to VGPRs)
    while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != plan.epoch)
      __builtin_amdgcn_s_sleep(4);
    const unsigned base = (unsigned)(((rank - 1) * NV4) * 512 + tid) * 16u;
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int f = 0; f < 2; ++f)
          acc[a][p][f] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_slab, (int)(base + (unsigned)((a * 2 + p) * 2 + f) * 8192u), 0, 16 /* sc1 */));
    // waited for HERE (as the bias in the epilogue): hipcc must not carry a vmcnt(0) for these loads into the K loop
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int p = 0; p < 2; ++p) asm volatile("" : "+v"(acc[a][p][0]), "+v"(acc[a][p][1]));
  };

  int cbuf = 0, ckt = seg0_k0, cke = seg0_k1, ctile = seg0_t, cti = 0;
  in_loop = true;
  for (int g = 0; g < total; ++g) {
    const unsigned boff = (unsigned)(cbuf * C::BUF);
    const unsigned xb = x_lane + boff, wb = w_lane + boff;
    if constexpr (CFG == 0) {
      // phase 0: quadrant (M0, N0); X1 of step g + 1
      read_x(std::integral_constant<int, C::X0>{}, xb);
      read_w(std::integral_constant<int, C::W0>{}, wb);
      stage(I1{});
      bar();
      pp_wait12(xf, wf);
      mfma16(I0{}, I0{});
      bar_load();
      // phase 1: (M0, N1); W0 of step g + 1
      read_w(std::integral_constant<int, C::W1>{}, wb);
      stage(I2{});
      bar();
      pp_wait4(wf);
      mfma16(I0{}, I1{});
      bar_load();
      // phase 2: (M1, N1); X0 of step g + 2
      read_x(std::integral_constant<int, C::X1>{}, xb);
      stage(I0{});
      bar();
      pp_wait8(xf);
      mfma16(std::integral_constant<int, 4>{}, I1{});
      bar_load();
      // phase 3: (M1, N0); W1 of step g + 2.  Everything of step g + 1 has landed when at most the LDS-DMAs issued after its
      // last half-tile (W0, phase 1) are in flight: X0, W1 of step g + 2 = 4
      read_w(std::integral_constant<int, C::W0>{}, wb);
      stage(I3{});
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      bar();
      pp_wait4(wf);
      mfma16(std::integral_constant<int, 4>{}, I0{});
      bar_load();
    } else {
      // phase 0: (M, N0); X1 of step g + 2
      read_x(std::integral_constant<int, 0>{}, xb);
      read_w(std::integral_constant<int, C::W0>{}, wb);
      stage(I1{});
      bar();
      pp_wait12(xf, wf);
      mfma16(I0{}, I0{});
      bar_load();
      // phase 1: (M, N1); W of step g + 2, X0 of step g + 3.  Step g + 1 has landed when at most the LDS-DMAs issued after ITS
      // W are in flight: X0(g+2), X1(g+2), W(g+2), X0(g+3) = 8
      read_w(std::integral_constant<int, C::W0 + 32 * 128>{}, wb);
      stage(I2{});
      stage(I0{});
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      bar();
      pp_wait4(wf);
      mfma16(I0{}, I1{});
      bar_load();
    }
    cbuf = cbuf + 1 == NBUF ? 0 : cbuf + 1;
    if constexpr (SK) {
      // this K step's counted wait is behind us: two K steps after the slab stores they are older than every operation the wait
      // may leave in flight (cfg 0: 8 LDS-DMAs per K step, vmcnt(4); cfg 1: 6 per K step, vmcnt(8)), i.e. retired = written through
      if (__builtin_expect(pub == 1, 0)) slab_publish();
      pub = pub > 0 ? pub - 1 : 0;
    }
    if (++ckt == cke) {
      if (AS_PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      if (SK && cke < nk) {                                   // an open tile: hand the accumulators to the next workgroup
        slab_store();
        pub = 2;
      } else {
        epilogue(ctile);
      }
      ++cti;
      ckt = 0;
      if (cti < nseg) {
        const Seg sg = seg_at(cti);
        ctile = sg.tile; ckt = sg.k0; cke = sg.k1;
      }
      if (SK && ckt > 0) slab_load();
      else zero_acc();
      if (AS_PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the wrapped tail of the operand stream lands before the LDS is freed
  if constexpr (SK) {
    if (pub > 0) slab_publish();                              // (a last segment shorter than two K steps)
  }
  if (grp == 0) bar();                                        // pairs with group 1's last barrier
#endif
}

// ---- stream-K workspace: one per (device, stream), made on first use, never freed (64 MiB of slabs + 8 KiB of flags) ----
// Launches on one stream are ordered, so a stream's slabs are reused launch after launch; the flags are never reset: each
// launch compares them with its own epoch.
struct PPWorkspace {
  void* slabs = nullptr;
  unsigned* flags = nullptr;
  unsigned epoch = 0;
};
constexpr size_t PP_SLAB_BYTES = (size_t)256 * 32 * 512 * 16;   // G <= 256 workgroups x the 256 x 256 tile's 32 vectors per thread
constexpr size_t PP_FLAG_BYTES = (size_t)256 * 8 * sizeof(unsigned);

PPWorkspace* pp_workspace(hipStream_t s) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, PPWorkspace> all;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  auto it = all.find({dev, s});
  if (it != all.end()) return it->second.slabs ? &it->second : nullptr;
  PPWorkspace ws;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
  void* mem = nullptr;
  if (hipMalloc(&mem, PP_SLAB_BYTES + PP_FLAG_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    all[{dev, s}] = ws;                                       // (remembered: this stream stays with whole tiles)
    return nullptr;
  }
  ws.slabs = mem;
  ws.flags = reinterpret_cast<unsigned*>(static_cast<char*>(mem) + PP_SLAB_BYTES);
  (void)hipMemsetAsync(ws.flags, 0, PP_FLAG_BYTES, s);
  return &(all[{dev, s}] = ws);
}

// AS_GEMM_PP_SK = 0: whole tiles only | 1: the stream-K tail wherever it applies | unset: where the round model below predicts a gain
// (re-read per call under AS_GEMM_PP_DYN).  Measured on one MI355X (profiles/r06_gemm_pp_sk.md): the hand-over costs ~9 us per launch
// with 128 KiB slabs (256 x 128 tiles) and ~20 us with 256 KiB slabs -- the write-through slab stores retire in order with the LDS-DMAs,
// so the K loop's counted waits stand behind them -- which eats the saved fraction of a round on the ViT-B shapes (fc1 792 tiles: 50.5
// vs 50.8 us) and pays where the last round is mostly empty (ViT-L fc2, 328 tiles of 64 K steps: 103.8 -> 89.2 us).
int pp_sk_mode() {
  static const bool dyn = getenv("AS_GEMM_PP_DYN") != nullptr;
  static const char* e0 = getenv("AS_GEMM_PP_SK");
  const char* e = dyn ? getenv("AS_GEMM_PP_SK") : e0;
  return e == nullptr ? -1 : e[0] == '0' ? 0 : 1;
}
bool pp_sk_pays(int cfg, int tiles, int grid, int nk, bool gelu) {
  const float tk = cfg == 0 ? 1.55f : 0.85f, epi = (cfg == 0 ? 3.0f : 2.0f) + (gelu ? (cfg == 0 ? 5.0f : 2.5f) : 0.0f);
  const float rounds = (float)as_ceil_div(tiles, grid);
  const float whole = rounds * (nk * tk + epi);
  const float shared = (float)tiles * nk / grid * tk + rounds * epi + (cfg == 0 ? 20.0f : 9.0f);
  return shared < 0.92f * whole;
}

template <int CFG, int EM, int ACT>
int launch_pp(const void* A, const void* W, const float* bias, void* out, int M, int Nout, int K, PPEpi epi, hipStream_t s) {
  using C = PPCfg<CFG>;
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? n - n % 8 : 8;
  }();
  const int tiles = as_ceil_div(M, C::BM) * (Nout / C::BN);
  const int grid = tiles >= cus ? cus : as_round_up(tiles, 8);
  const size_t lds = (size_t)C::NBUF * C::BUF;
  // stream-K tail: more than one round and a ragged last one (GELU' form excluded: its epilogue loads beside the slab loads were
  // not worth a second variant).  All `grid` workgroups must be co-resident for the hand-over to be prompt (one per CU: they are)
  PPPlan plan{0, 0, 0u, nullptr, nullptr};
  bool sk = false;
  if (tiles > grid && tiles % grid != 0 && grid <= 256 && !(EM == 0 && ACT == 3) && (long long)2 * grid * (K / 64) * grid < (1LL << 31) &&
      (pp_sk_mode() == 1 || (pp_sk_mode() < 0 && pp_sk_pays(CFG, tiles, grid, K / 64, EM != 1 && (ACT == 1 || ACT == 2))))) {
    if (PPWorkspace* ws = pp_workspace(s)) {
      plan.D = tiles / grid - 1;
      plan.sk_tiles = tiles - plan.D * grid;
      if (++ws->epoch == 0u) ws->epoch = 1u;
      plan.epoch = ws->epoch;
      plan.slabs = ws->slabs;
      plan.flags = ws->flags;
      sk = true;
    }
  }
  static std::atomic<bool> attr_set[2] = {{false}, {false}};
  if (!attr_set[sk]) {
    if (sk) (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<CFG, EM, ACT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<CFG, EM, ACT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set[sk] = true;
  }
  if (sk)
    hipLaunchKernelGGL((gemm_pp_kernel<CFG, EM, ACT, true>), dim3(grid), dim3(512), lds, s, (const __bf16*)A, (const __bf16*)W, bias,
                       (__bf16*)out, M, Nout, K, epi, plan);
  else
    hipLaunchKernelGGL((gemm_pp_kernel<CFG, EM, ACT, false>), dim3(grid), dim3(512), lds, s, (const __bf16*)A, (const __bf16*)W, bias,
                       (__bf16*)out, M, Nout, K, epi, plan);
  AS_CHECK_LAUNCH("gemm_pp");
  return AS_OK;
}

}  // namespace

// ---- internal entry points (gemm.hip dispatches here; not part of the C ABI) ----
// cfg: 0 = 256 x 256 tiles, 1 = 256 x 128.  Preconditions (checked by the callers through as_pp_applies): bf16,
// K % 64 == 0, Nout % BN == 0, 16-byte aligned rows.
bool as_pp_applies(int M, int Nout, int K, int cfg) {
  const int bn = cfg == 0 ? 256 : 128;
  return M >= 256 && K >= 128 && K % 64 == 0 && Nout % bn == 0 && (long long)M * K < (1LL << 30) && (long long)Nout * K < (1LL << 30);
}
// act: 0 none, 1 GELU, 4 ReLU; 2 = training fc1 (pre-activation -> `pre`, its GELU -> out); 3 = out = acc * GELU'(pre)
int as_pp_linear(const void* x, const void* W, const float* bias, void* out, void* pre, int M, int Nout, int K, int act, int cfg,
                 hipStream_t s) {
  PPEpi epi{pre, nullptr, nullptr, 0, 0, 0, 0};
#define PP_GO(C_)                                                                           \
  switch (act) {                                                                            \
    case 1: return launch_pp<C_, 0, 1>(x, W, bias, out, M, Nout, K, epi, s);                \
    case 2: return launch_pp<C_, 0, 2>(x, W, bias, out, M, Nout, K, epi, s);                \
    case 3: return launch_pp<C_, 0, 3>(x, W, bias, out, M, Nout, K, epi, s);                \
    case 4: return launch_pp<C_, 0, 4>(x, W, bias, out, M, Nout, K, epi, s);                \
    default: return launch_pp<C_, 0, 0>(x, W, bias, out, M, Nout, K, epi, s);               \
  }
  if (cfg == 0) { PP_GO(0) }
  PP_GO(1)
#undef PP_GO
}
int as_pp_deconv(const void* x, const void* W4, const float* bias4, void* out, int M, int w, int cin, int cout, int act, int cfg,
                 hipStream_t s) {
  PPEpi epi{nullptr, nullptr, nullptr, w, 0, cout, 0};
  if (cfg == 0) return act == 1 ? launch_pp<0, 2, 1>(x, W4, bias4, out, M, 4 * cout, cin, epi, s)
                                : launch_pp<0, 2, 0>(x, W4, bias4, out, M, 4 * cout, cin, epi, s);
  return act == 1 ? launch_pp<1, 2, 1>(x, W4, bias4, out, M, 4 * cout, cin, epi, s)
                  : launch_pp<1, 2, 0>(x, W4, bias4, out, M, 4 * cout, cin, epi, s);
}
int as_pp_qkv(const void* x, const void* Wqkv, const float* bqkv, void* q, void* k, void* vt, int B, int N, int Npad, int D, int h,
              hipStream_t s) {
  PPEpi epi{q, k, vt, N, Npad, D, h};
  return launch_pp<1, 1, 0>(x, Wqkv, bqkv, nullptr, B * N, 3 * D, D, epi, s);
}
