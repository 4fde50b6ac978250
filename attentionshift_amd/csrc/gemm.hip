// Linear / QKV projection GEMM for gfx950:  C[M,Nout] = act(A[M,K] . W[Nout,K]^T + bias)
//
// Replaces nn.Linear in Attention / Mlp (reference models/vision_transformer.py:47-59, 75-77, 84).
// 128x128x32 workgroup tile, 4 waves (2x2), each wave 2x2 MFMA 32x32 tiles (64 accumulator VGPRs);
// both operands are K-contiguous so A and W fragments are 16-byte (bf16) row reads; LDS rows are padded
// by 16 B so the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots (conflict-free).
// The QKV epilogue writes k as [B,h,Npad,64], q in the fragment-major tile layout of common.h (qf_elem) and V
// TRANSPOSED as [B,h,64,Npad]; V tiles are computed with swapped MFMA operands so lanes run along tokens and
// the transposed store stays coalesced.
#include "common.h"

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
};

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

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
              dst[at] = from_f32<T>(acc[i][j][r] + bv);
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
// bf16 fast path: 128x128x64 tiles, operands streamed global -> LDS with global_load_lds_dwordx4 (no VGPR
// round trip), two LDS buffers (64 KiB), one barrier per K step.  The LDS image of a tile is lane-linear
// ([row][8 x 16-B chunks], 1 KiB per wave instruction), so the bank-conflict swizzle is applied on the SOURCE
// address: LDS chunk c of row r holds global chunk c ^ (r & 7); fragment reads apply the same XOR (guide rule 21).
// ---------------------------------------------------------------------------------------------------------
constexpr int GK = 64;                       // K step (elements)
constexpr int G_TILE_BYTES = BM * GK * 2;    // 16 KiB per operand tile

template <int MODE>
__global__ __launch_bounds__(NT) void gemm_glds_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W,
                                                       const float* __restrict__ bias, __bf16* __restrict__ out, int M,
                                                       int Nout, int K, int act, QkvEpi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 buffers][A tile | W tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  // loader: wave w issues 4 + 4 one-KiB pieces per K step; piece j covers tile rows (w*4+j)*8 .. +7
  const int lr = lane >> 3, lc = lane & 7;
  const char* srcA[4];
  const char* srcW[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 8 + lr;
    const int cs = lc ^ (r & 7);
    srcA[j] = reinterpret_cast<const char*>(A + (size_t)min(m0 + r, M - 1) * K) + cs * 16;
    srcW[j] = reinterpret_cast<const char*>(W + (size_t)min(n0 + r, Nout - 1) * K) + cs * 16;
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * (2 * G_TILE_BYTES);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = (wave * 4 + j) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + (size_t)kt * GK * 2),
                                       (__attribute__((address_space(3))) void*)(base + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[j] + (size_t)kt * GK * 2),
                                       (__attribute__((address_space(3))) void*)(base + G_TILE_BYTES + piece), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  bool swapped[2] = {false, false};
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) swapped[j] = (n0 + wn * 64 + j * 32) >= 2 * epi.D;
  }

  const int nk = K / GK;
  stage(0, 0);
  __syncthreads();                                   // drains the LDS-DMA (vmcnt(0)) and publishes the tile
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* As = smem + (kt & 1) * (2 * G_TILE_BYTES);
    const char* Bs = As + G_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks) {
      Frag<__bf16> fa[2], fb[2];
      const int g = ks * 2 + half;                   // 16-byte chunk of the 128-byte row
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + li;
        fa[i].load16B(reinterpret_cast<const __bf16*>(As + ra * 128 + ((g ^ (ra & 7)) << 4)));
        const int rb = wn * 64 + i * 32 + li;
        fb[i].load16B(reinterpret_cast<const __bf16*>(Bs + rb * 128 + ((g ^ (rb & 7)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (MODE == 1 && swapped[j]) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
          else acc[i][j] = mma32(fa[i], fb[j], acc[i][j]);
        }
    }
    __syncthreads();                                 // next tile landed, everyone finished reading this one
  }
  gemm_epilogue<__bf16, MODE>(acc, swapped, bias, out, M, Nout, act, epi, m0, n0, wm, wn, li, half);
}

template <int MODE>
int launch_gemm_glds(const void* A, const void* W, const float* bias, void* out, int M, int Nout, int K, int act,
                     QkvEpi epi, hipStream_t s) {
  dim3 grid(as_ceil_div(M, BM), as_ceil_div(Nout, BN));
  const size_t lds = 4 * (size_t)G_TILE_BYTES;      // 64 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_glds_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_glds_kernel<MODE>), grid, dim3(NT), lds, s, (const __bf16*)A, (const __bf16*)W, bias,
                     (__bf16*)out, M, Nout, K, act, epi);
  AS_CHECK_LAUNCH("gemm_glds");
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

}  // namespace

extern "C" int as_npad(int N) { return as_round_up(N, 64); }

extern "C" int as_linear_fwd(const void* x, const void* W, const float* bias, void* out, int M, int Nout, int K,
                             int dtype, int act, as_stream_t stream) {
  AS_REQUIRE(x && W && out, AS_E_BADARG, "as_linear_fwd: null pointer");
  AS_REQUIRE(M > 0 && Nout > 0 && K > 0 && K % BK == 0, AS_E_BADARG, "as_linear_fwd: need M,N>0 and K %% 32 == 0 (K=%d)", K);
  AS_REQUIRE(act == 0 || act == 1, AS_E_BADARG, "as_linear_fwd: act must be 0 or 1");
  QkvEpi epi{};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16 && K % GK == 0) return launch_gemm_glds<0>(x, W, bias, out, M, Nout, K, act, epi, s);
  if (dtype == AS_BF16) return launch_gemm<__bf16, 0>(x, W, bias, out, M, Nout, K, act, epi, s);
  if (dtype == AS_F32) return launch_gemm<float, 0>(x, W, bias, out, M, Nout, K, act, epi, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_linear_fwd: dtype %d", dtype);
}

extern "C" int as_qkv_fwd(const void* x, const void* Wqkv, const float* bqkv, void* q, void* k, void* vt, int B,
                          int N, int D, int h, int dtype, as_stream_t stream) {
  AS_REQUIRE(x && Wqkv && q && k && vt, AS_E_BADARG, "as_qkv_fwd: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && h > 0 && D == h * AS_HEAD_DIM, AS_E_UNSUPPORTED,
             "as_qkv_fwd: head dim must be 64 (D=%d h=%d)", D, h);
  QkvEpi epi{q, k, vt, N, as_npad(N), D, h};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16 && D % GK == 0) return launch_gemm_glds<1>(x, Wqkv, bqkv, nullptr, B * N, 3 * D, D, 0, epi, s);
  if (dtype == AS_BF16) return launch_gemm<__bf16, 1>(x, Wqkv, bqkv, nullptr, B * N, 3 * D, D, 0, epi, s);
  if (dtype == AS_F32) return launch_gemm<float, 1>(x, Wqkv, bqkv, nullptr, B * N, 3 * D, D, 0, epi, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_qkv_fwd: dtype %d", dtype);
}
