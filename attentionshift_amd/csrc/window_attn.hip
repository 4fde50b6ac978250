// Swin window attention (BASELINE config 5) as ONE kernel over the un-partitioned token grid.
//
// Replaces, for one SwinTransformerBlock (reference models/swin_transformer.py):
//   F.pad to a multiple of the window (:271-276), torch.roll cyclic shift (:279-280), window_partition (:292-293),
//   WindowAttention.forward core (:131-153: q*scale, q k^T, + relative_position_bias_table[index], + shift mask,
//   softmax, attn @ v), window_reverse (:299-300), the reverse roll (:303-304) and the un-pad slice (:308-309).
// Because nn.Linear acts per token, the QKV GEMM runs on the original [B,H,W,C] grid WITHOUT its bias; this kernel
// adds the bias while gathering each window's 49 tokens through the shift / partition index map, and uses the bias
// alone for padded tokens (the reference pads AFTER norm1, so a padded token's qkv is exactly the bias).  The shift
// mask (create_attn_mask :233-256) is evaluated from the 3x3 region ids instead of being read from a [nW,49,49] table,
// and the relative-position bias is gathered from the [(2w-1)^2, h] table through the closed-form index (:120-130).
// Results are scattered straight back to the original token positions: no pad / roll / partition / reverse copies.
//
// One wave per (window, head): lane i owns query row i (49 of 64 lanes), K and V of the window live in LDS as fp32 and
// are read as broadcasts; all arithmetic is fp32 (the softmax in exp(x - max) / sum form, as torch's).
#include "common.h"

namespace {

template <typename T, int WS, int HD>
__global__ __launch_bounds__(64) void window_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bqkv,
                                                             const float* __restrict__ table, T* __restrict__ out,
                                                             float* __restrict__ attn_out, int B, int H, int W, int h,
                                                             int shift) {
  constexpr int N = WS * WS, TB = (2 * WS - 1) * (2 * WS - 1);
  __shared__ __attribute__((aligned(16))) float Ks[N][HD];
  __shared__ __attribute__((aligned(16))) float Vs[N][HD];
  __shared__ float Ss[64 * N];
  __shared__ float tab[TB];
  __shared__ int rid[64];
  const int C = h * HD;
  const int nWh = as_ceil_div_dev(H, WS), nWw = as_ceil_div_dev(W, WS);
  const int Hp = nWh * WS, Wp = nWw * WS;
  const int win = blockIdx.x, head = blockIdx.y;
  const int b = win / (nWh * nWw), wi = (win / nWw) % nWh, wj = win % nWw;
  const int lane = threadIdx.x;
  const int li = min(lane, N - 1);
  const int a = li / WS, c_ = li % WS;
  const int hs = wi * WS + a, ws_ = wj * WS + c_;                  // coordinates in the shifted, padded grid
  const int ho = (hs + shift) % Hp, wo = (ws_ + shift) % Wp;      // torch.roll(x, -shift): shifted[i] = x[(i + shift) % Hp]
  const bool real = ho < H && wo < W;
  const T* tok = qkv + (((size_t)b * H + (real ? ho : 0)) * W + (real ? wo : 0)) * (size_t)(3 * C) + head * HD;

  float q[HD];
  const float scale = rsqrtf((float)HD);
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    const float bq = bqkv ? bqkv[head * HD + c] : 0.0f, bk = bqkv ? bqkv[C + head * HD + c] : 0.0f,
                bv = bqkv ? bqkv[2 * C + head * HD + c] : 0.0f;
    q[c] = ((real ? to_f32<T>(tok[c]) : 0.0f) + bq) * scale;
    if (lane < N) {
      Ks[li][c] = (real ? to_f32<T>(tok[C + c]) : 0.0f) + bk;
      Vs[li][c] = (real ? to_f32<T>(tok[2 * C + c]) : 0.0f) + bv;
    }
  }
  for (int t = lane; t < TB; t += 64) tab[t] = table[(size_t)t * h + head];
  {
    const int rh = hs < Hp - WS ? 0 : (hs < Hp - shift ? 1 : 2), rw = ws_ < Wp - WS ? 0 : (ws_ < Wp - shift ? 1 : 2);
    rid[lane] = shift > 0 ? 3 * rh + rw : 0;
  }
  __syncthreads();

  // scores of this lane's row live in LDS (row stride N = 49 floats is odd: conflict-free across lanes)
  float* srow = &Ss[lane * N];
  const int myrid = rid[li];
  float m = -INFINITY;
  for (int j = 0, aj = 0, cj = 0; j < N; ++j) {
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(&Ks[j][c]);
      acc = fmaf(q[c], kv.x, acc);
      acc = fmaf(q[c + 1], kv.y, acc);
      acc = fmaf(q[c + 2], kv.z, acc);
      acc = fmaf(q[c + 3], kv.w, acc);
    }
    acc += tab[(a - aj + WS - 1) * (2 * WS - 1) + (c_ - cj + WS - 1)];
    if (shift > 0 && rid[j] != myrid) acc += -100.0f;
    srow[j] = acc;
    m = fmaxf(m, acc);
    if (++cj == WS) { cj = 0; ++aj; }
  }
  float sum = 0.0f;
  for (int j = 0; j < N; ++j) {
    const float e = expf(srow[j] - m);
    srow[j] = e;
    sum += e;
  }
  const float inv = 1.0f / sum;
  float o[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) o[c] = 0.0f;
  for (int j = 0; j < N; ++j) {
    const float p = srow[j] * inv;
    srow[j] = p;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 vv = *reinterpret_cast<const float4*>(&Vs[j][c]);
      o[c] = fmaf(p, vv.x, o[c]);
      o[c + 1] = fmaf(p, vv.y, o[c + 1]);
      o[c + 2] = fmaf(p, vv.z, o[c + 2]);
      o[c + 3] = fmaf(p, vv.w, o[c + 3]);
    }
  }
  if (lane < N) {
    if (attn_out) {
      float* row = attn_out + (((size_t)win * h + head) * N + lane) * N;
      for (int j = 0; j < N; ++j) row[j] = srow[j];
    }
    if (real) {
      T* dst = out + (((size_t)b * H + ho) * W + wo) * (size_t)C + head * HD;
#pragma unroll
      for (int c = 0; c < HD; ++c) dst[c] = from_f32<T>(o[c]);
    }
  }
}

template <typename T>
int launch_window_attn(const void* qkv, const float* bqkv, const float* table, void* out, float* attn_out, int B, int H,
                       int W, int h, int shift, hipStream_t s) {
  const int nW = as_ceil_div(H, 7) * as_ceil_div(W, 7);
  hipLaunchKernelGGL((window_attn_fwd_kernel<T, 7, 32>), dim3(B * nW, h), dim3(64), 0, s, (const T*)qkv, bqkv, table,
                     (T*)out, attn_out, B, H, W, h, shift);
  AS_CHECK_LAUNCH("window_attn_fwd");
  return AS_OK;
}

}  // namespace

extern "C" int as_window_attn_fwd(const void* qkv, const float* bqkv, const float* table, void* out, float* attn_out,
                                  int B, int H, int W, int C, int h, int ws, int shift, int dtype, as_stream_t stream) {
  AS_REQUIRE(qkv && table && out, AS_E_BADARG, "as_window_attn_fwd: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && h > 0, AS_E_BADARG, "as_window_attn_fwd: bad sizes");
  AS_REQUIRE(ws == 7 && C == h * 32, AS_E_UNSUPPORTED,
             "as_window_attn_fwd: window 7 and head dim 32 only (every Swin variant of the reference) (ws=%d C=%d h=%d)",
             ws, C, h);
  AS_REQUIRE(shift >= 0 && shift < ws, AS_E_BADARG, "as_window_attn_fwd: need 0 <= shift < ws");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_window_attn<__bf16>(qkv, bqkv, table, out, attn_out, B, H, W, h, shift, s);
  if (dtype == AS_F32) return launch_window_attn<float>(qkv, bqkv, table, out, attn_out, B, H, W, h, shift, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_window_attn_fwd: dtype %d", dtype);
}
