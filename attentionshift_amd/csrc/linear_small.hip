// Small-M fp32 linear for gfx950:  out[M, Nout] = act(x[M, K] . W[Nout, K]^T + bias), M of a few hundred rows.
//
// The point head of VisionTransformerDet (reference mmdet/models/backbones/visual_transformer_det.py:26-38, 145-146,
// 262-267: class_embed / bbox_embed, two 3-layer FFNs over the B * T = 200 point tokens, fp32) was the last vendor-library
// GEMM inside the headline step: the 128 x 128 tiles of gemm.hip's fp32 kernel put 200 rows on 24 of the 256 CUs (168 us per
// layer), so round 5 sent these five products to hipBLASLt (0.087 ms per step).  This kernel fills the chip from the other
// side: a workgroup owns a 32-row x 64-column output tile and its FOUR WAVES SPLIT K (a quarter each, straight from global
// memory -- both operands are K-contiguous, every lane reads 32 bytes of its row per step), the partial tiles meet in LDS in
// fixed wave order (deterministic), and bias / activation ride in the reduction.  M = 200, Nout = 1536, K = 768: 7 x 24 =
// 168 workgroups = 672 waves for 1024 SIMDs; the arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 chains, common.h mma32).
// Row strides are arguments, so a column slice of a packed activation (the two heads share their first layer) is read in
// place.
#include "common.h"

namespace {

constexpr int LS_BM = 32, LS_BN = 64, LS_NT = 256;

// act: 0 none, 1 exact erf GELU, 4 ReLU, 5 sigmoid
template <int ACT> __device__ __forceinline__ float ls_act(float v) {
  if constexpr (ACT == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  else if constexpr (ACT == 4) return fmaxf(v, 0.0f);
  else if constexpr (ACT == 5) return 1.0f / (1.0f + expf(-v));
  else return v;
}

template <int ACT>
__global__ __launch_bounds__(LS_NT) void linear_small_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                             int M, int Nout, int K) {
  __shared__ float part[4][LS_BM][LS_BN + 1];                 // [wave = K quarter][row][col] (+1: the reduction reads columns)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * LS_BM, n0 = blockIdx.y * LS_BN;
  // K range of this wave: quarters in units of 16 (one mma32 step), the remainder to the last wave
  const int steps = K / 16, per = steps / 4;
  const int s_beg = wave * per, s_end = wave == 3 ? steps : s_beg + per;
  const float* pa = x + (size_t)min(m0 + li, M - 1) * ldx + 8 * half;
  const float* pb0 = W + (size_t)min(n0 + li, Nout - 1) * K + 8 * half;
  const float* pb1 = W + (size_t)min(n0 + 32 + li, Nout - 1) * K + 8 * half;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
#pragma unroll 2
  for (int s = s_beg; s < s_end; ++s) {
    Frag<float> fa, fb0, fb1;
    fa.load16B(pa + 16 * s);
    fb0.load16B(pb0 + 16 * s);
    fb1.load16B(pb1 + 16 * s);
    acc0 = mma32(fa, fb0, acc0);                              // D[token i][feature j]
    acc1 = mma32(fa, fb1, acc1);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    part[wave][acc_row(r, half)][li] = acc0[r];
    part[wave][acc_row(r, half)][32 + li] = acc1[r];
  }
  __syncthreads();
  // 2048 outputs, 8 per thread: thread -> (row = tid / 8 .. , 8 consecutive columns); partials summed in wave order
  const int row = tid >> 3, c0 = (tid & 7) * 8;
  if (m0 + row >= M) return;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int col = n0 + c0 + c;
    if (col >= Nout) break;
    float v = ((part[0][row][c0 + c] + part[1][row][c0 + c]) + part[2][row][c0 + c]) + part[3][row][c0 + c];
    if (bias != nullptr) v += bias[col];
    out[(size_t)(m0 + row) * ldo + col] = ls_act<ACT>(v);
  }
}

}  // namespace

extern "C" int as_linear_small_fwd(const float* x, int ldx, const float* W, const float* bias, float* out, int ldo, int M, int Nout,
                                   int K, int act, as_stream_t stream) {
  AS_REQUIRE(x && W && out, AS_E_BADARG, "as_linear_small_fwd: null pointer");
  AS_REQUIRE(M > 0 && Nout > 0 && K >= 64 && K % 16 == 0 && ldx >= K && ldo >= Nout && ldx % 4 == 0, AS_E_BADARG,
             "as_linear_small_fwd: need M, Nout > 0, K %% 16 == 0, K >= 64, ldx >= K (ldx %% 4 == 0), ldo >= Nout (M=%d N=%d K=%d ldx=%d ldo=%d)",
             M, Nout, K, ldx, ldo);
  AS_REQUIRE(act == 0 || act == 1 || act == 4 || act == 5, AS_E_BADARG, "as_linear_small_fwd: act must be 0, 1 (GELU), 4 (ReLU) or 5 (sigmoid)");
  AS_REQUIRE(((size_t)x % 16 == 0) && ((size_t)W % 16 == 0) && K % 4 == 0, AS_E_BADARG, "as_linear_small_fwd: operands must be 16-byte aligned");
  const dim3 grid(as_ceil_div(M, LS_BM), as_ceil_div(Nout, LS_BN));
  hipStream_t s = (hipStream_t)stream;
  switch (act) {
    case 1: hipLaunchKernelGGL(linear_small_kernel<1>, grid, dim3(LS_NT), 0, s, x, ldx, W, bias, out, ldo, M, Nout, K); break;
    case 4: hipLaunchKernelGGL(linear_small_kernel<4>, grid, dim3(LS_NT), 0, s, x, ldx, W, bias, out, ldo, M, Nout, K); break;
    case 5: hipLaunchKernelGGL(linear_small_kernel<5>, grid, dim3(LS_NT), 0, s, x, ldx, W, bias, out, ldo, M, Nout, K); break;
    default: hipLaunchKernelGGL(linear_small_kernel<0>, grid, dim3(LS_NT), 0, s, x, ldx, W, bias, out, ldo, M, Nout, K); break;
  }
  AS_CHECK_LAUNCH("linear_small");
  return AS_OK;
}
