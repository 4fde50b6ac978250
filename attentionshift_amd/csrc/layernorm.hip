// Fused residual add + LayerNorm for the pre-LN transformer block (reference models/vision_transformer.py:109-124):
//     x_out = x_in + delta                    (fp32 residual stream; delta = the previous sub-layer's output, T)
//     y_out = LayerNorm(x_out) * gamma + beta (written in the GEMM input dtype T)
// One pass over HBM (read x_in, delta; write x_out, y_out) instead of the four elementwise launches it replaces
// (cast, add, layer_norm, cast: 208 -> 78 MB per call at ViT-B / 1024^2 / B=2).  One wave per row, the row lives in
// registers; mean and variance are the two-pass fp32 forms (sum, then sum of squared deviations).
#include "common.h"

namespace {

template <typename T, int VPL>   // VPL = float4 groups per lane: D <= 64 * 4 * VPL
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x_in, const T* __restrict__ delta,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ x_out, T* __restrict__ y_out, int M, int D,
                                                            float eps, const float* __restrict__ dscale, int rps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const size_t base = (size_t)row * D;
  const float ds = dscale != nullptr ? dscale[row / rps] : 1.0f;    // per-sample scale of delta (stochastic depth)
  float v[VPL][4];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const float4 a = *reinterpret_cast<const float4*>(x_in + base + c);
      v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
      if (delta != nullptr) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[i][t] = dscale != nullptr ? fmaf(ds, to_f32<T>(delta[base + c + t]), v[i][t])
                                                                     : v[i][t] + to_f32<T>(delta[base + c + t]);
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.0f;
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float d = v[i][t] - mean;
        q = fmaf(d, d, q);
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      if (x_out != nullptr) *reinterpret_cast<float4*>(x_out + base + c) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
      if (y_out != nullptr) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 b = *reinterpret_cast<const float4*>(beta + c);
        const float y0 = (v[i][0] - mean) * rstd * g.x + b.x, y1 = (v[i][1] - mean) * rstd * g.y + b.y;
        const float y2 = (v[i][2] - mean) * rstd * g.z + b.z, y3 = (v[i][3] - mean) * rstd * g.w + b.w;
        T* yp = y_out + base + c;
        yp[0] = from_f32<T>(y0); yp[1] = from_f32<T>(y1); yp[2] = from_f32<T>(y2); yp[3] = from_f32<T>(y3);
      }
    }
  }
}

template <typename T>
int launch_add_ln(const float* x_in, const void* delta, const float* gamma, const float* beta, float* x_out, void* y_out,
                  int M, int D, float eps, const float* dscale, int rps, hipStream_t s) {
  const int vpl = as_ceil_div(D, 256);
  dim3 grid(as_ceil_div(M, 4));
#define AS_LN(V)                                                                                                   \
  hipLaunchKernelGGL((add_layernorm_kernel<T, V>), grid, dim3(256), 0, s, x_in, (const T*)delta, gamma, beta, x_out, \
                     (T*)y_out, M, D, eps, dscale, rps)
  switch (vpl) {
    case 1: AS_LN(1); break;
    case 2: AS_LN(2); break;
    case 3: AS_LN(3); break;
    case 4: AS_LN(4); break;
    case 5: AS_LN(5); break;
    case 6: AS_LN(6); break;
    case 7: AS_LN(7); break;
    case 8: AS_LN(8); break;
    case 9: case 10: AS_LN(10); break;
    case 11: case 12: AS_LN(12); break;
    default: AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_add_layernorm: D=%d (max 3072)", D);
  }
#undef AS_LN
  AS_CHECK_LAUNCH("add_layernorm");
  return AS_OK;
}

}  // namespace

// =====================================================================================================
// Backward of the fused residual add + LayerNorm (trainable path).  With x = x_in + delta (saved, fp32) and
// y = LN(x) * gamma + beta:
//     g = dy * gamma,  xhat = (x - mean) * rstd
//     dx = dx_res + rstd * (g - mean(g) - xhat * mean(g * xhat))         (dx_res = the residual stream's own gradient)
// dx is the gradient of BOTH x_in (fp32, dx_out) and delta (T, ddelta_out).  mean / rstd are recomputed from the row in
// registers (two-pass, as the forward).  dgamma / dbeta: every workgroup walks its rows in order and leaves ONE partial
// [2, D]; a second launch adds the partials in workgroup order (deterministic, no atomics).
// =====================================================================================================
namespace {

template <typename T, int VPL>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                                const float* __restrict__ dx_res,
                                                                const float* __restrict__ gamma, float* __restrict__ dx_out,
                                                                T* __restrict__ ddelta_out, float* __restrict__ part, int M,
                                                                int D, float eps, const float* __restrict__ dscale, int rps) {
  __shared__ float red[2][4][VPL * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float dg[VPL][4], db[VPL][4], gm[VPL][4];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) { dg[i][t] = 0.0f; db[i][t] = 0.0f; gm[i][t] = (gamma != nullptr && c < D) ? gamma[c + t] : 1.0f; }
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const size_t base = (size_t)row * D;
    float v[VPL][4], g[VPL][4];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        const float4 a = *reinterpret_cast<const float4*>(x + base + c);
        v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
        if (dy == nullptr) {
#pragma unroll
          for (int t = 0; t < 4; ++t) g[i][t] = 0.0f;
        } else if constexpr (sizeof(T) == 2) {             // one 8-byte load of the four bf16 values
          const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(dy + base + c);
#pragma unroll
          for (int t = 0; t < 4; ++t) g[i][t] = (float)d4[t];
        } else {
          const float4 d4 = *reinterpret_cast<const float4*>(dy + base + c);
          g[i][0] = d4.x; g[i][1] = d4.y; g[i][2] = d4.z; g[i][3] = d4.w;
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) { v[i][t] = 0.0f; g[i][t] = 0.0f; }
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { const float d = v[i][t] - mean; q = fmaf(d, d, q); }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float sg = 0.0f, sgx = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float xh = (v[i][t] - mean) * rstd;
          dg[i][t] = fmaf(g[i][t], xh, dg[i][t]);                 // dgamma += dy * xhat
          db[i][t] += g[i][t];                                     // dbeta  += dy
          const float gg = g[i][t] * gm[i][t];
          g[i][t] = gg;
          v[i][t] = xh;
          sg += gg;
          sgx = fmaf(gg, xh, sgx);
        }
      }
    }
    const float mg = wave_sum(sg) / (float)D, mgx = wave_sum(sgx) / (float)D;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        float o[4];
        float4 r4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (dx_res != nullptr) r4 = *reinterpret_cast<const float4*>(dx_res + base + c);
        const float r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          o[t] = rstd * (g[i][t] - mg - v[i][t] * mgx);
          if (dx_res != nullptr) o[t] += r[t];
        }
        if (dx_out != nullptr) *reinterpret_cast<float4*>(dx_out + base + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (ddelta_out != nullptr) {
          const float ds = dscale != nullptr ? dscale[row / rps] : 1.0f;        // d(x + ds * delta) / d delta
          if constexpr (sizeof(T) == 2) {
            bf16x4 w4;
#pragma unroll
            for (int t = 0; t < 4; ++t) w4[t] = (__bf16)(ds * o[t]);
            *reinterpret_cast<bf16x4*>(ddelta_out + base + c) = w4;
          } else {
            *reinterpret_cast<float4*>(ddelta_out + base + c) = make_float4(ds * o[0], ds * o[1], ds * o[2], ds * o[3]);
          }
        }
      }
    }
  }
  // the four waves' column partials, added in wave order
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      red[0][wave][(i * 64 + lane) * 4 + t] = dg[i][t];
      red[1][wave][(i * 64 + lane) * 4 + t] = db[i][t];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    part[((size_t)blockIdx.x * 2 + 0) * D + c] = ((red[0][0][c] + red[0][1][c]) + red[0][2][c]) + red[0][3][c];
    part[((size_t)blockIdx.x * 2 + 1) * D + c] = ((red[1][0][c] + red[1][1][c]) + red[1][2][c]) + red[1][3][c];
  }
}

// dgamma / dbeta = the workgroup partials added in a FIXED order: 16 groups of a 64-column slab each add every 16th
// partial (four interleaved running sums), then the 16 group sums are added in group order.  grid (ceil(D / 64), 2), 1024 threads.
__global__ __launch_bounds__(1024) void add_layernorm_bwd_reduce_kernel(const float* __restrict__ part,
                                                                        float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                        int nblk, int D) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.x * 64 + cl, which = blockIdx.y;
  // four independent running sums per group (partials grp, grp + 16, ... dealt round-robin), so four loads are in
  // flight; combined in a fixed order
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  if (c < D) {
    int i = grp;
    for (; i + 48 < nblk; i += 64) {
      a0 += part[((size_t)i * 2 + which) * D + c];
      a1 += part[((size_t)(i + 16) * 2 + which) * D + c];
      a2 += part[((size_t)(i + 32) * 2 + which) * D + c];
      a3 += part[((size_t)(i + 48) * 2 + which) * D + c];
    }
    for (; i < nblk; i += 16) a0 += part[((size_t)i * 2 + which) * D + c];
  }
  red[grp][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp != 0 || c >= D) return;
  float t = 0.0f;
#pragma unroll
  for (int g = 0; g < 16; ++g) t += red[g][cl];
  float* dst = which == 0 ? dgamma : dbeta;
  if (dst != nullptr) dst[c] = t;
}

constexpr int LNB_BLOCKS = 1024;      // 4 workgroups (16 waves) per CU: the pass is a stream, it needs the loads in flight

template <typename T>
int launch_add_ln_bwd(const float* x, const void* dy, const float* dx_res, const float* gamma, float* dx_out, void* ddelta,
                      float* dgamma, float* dbeta, float* part, int M, int D, float eps, const float* dscale, int rps,
                      hipStream_t s) {
  const int vpl = as_ceil_div(D, 256);
  const int nblk = as_ceil_div(M, 4) < LNB_BLOCKS ? as_ceil_div(M, 4) : LNB_BLOCKS;
#define AS_LNB(V)                                                                                                    \
  hipLaunchKernelGGL((add_layernorm_bwd_kernel<T, V>), dim3(nblk), dim3(256), 0, s, x, (const T*)dy, dx_res, gamma, dx_out, \
                     (T*)ddelta, part, M, D, eps, dscale, rps)
  switch (vpl) {
    case 1: AS_LNB(1); break;
    case 2: AS_LNB(2); break;
    case 3: AS_LNB(3); break;
    case 4: AS_LNB(4); break;
    case 5: AS_LNB(5); break;                        // D = 1280 (ViT-H)
    case 6: AS_LNB(6); break;
    case 7: AS_LNB(7); break;
    case 8: AS_LNB(8); break;
    default: AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_add_layernorm_bwd: D=%d (max 2048)", D);
  }
#undef AS_LNB
  AS_CHECK_LAUNCH("add_layernorm_bwd");
  if (dgamma != nullptr || dbeta != nullptr) {
    hipLaunchKernelGGL(add_layernorm_bwd_reduce_kernel, dim3(as_ceil_div(D, 64), 2), dim3(1024), 0, s, (const float*)part,
                       dgamma, dbeta, nblk, D);
    AS_CHECK_LAUNCH("add_layernorm_bwd_reduce");
  }
  return AS_OK;
}

}  // namespace

extern "C" size_t as_add_layernorm_bwd_workspace_bytes(int M, int D) {
  if (M <= 0 || D <= 0) return 0;
  return (size_t)LNB_BLOCKS * 2 * D * sizeof(float);
}

extern "C" int as_add_layernorm_bwd(const float* x, const void* dy, const float* dx_res, const float* gamma, float eps,
                                    float* dx_out, void* ddelta_out, float* dgamma, float* dbeta, void* workspace,
                                    size_t workspace_bytes, int M, int D, int dtype, as_stream_t stream) {
  return as_add_layernorm_bwd_scaled(x, dy, dx_res, gamma, eps, dx_out, ddelta_out, dgamma, dbeta, workspace, workspace_bytes,
                                     M, D, dtype, nullptr, 1, stream);
}

extern "C" int as_add_layernorm_bwd_scaled(const float* x, const void* dy, const float* dx_res, const float* gamma, float eps,
                                           float* dx_out, void* ddelta_out, float* dgamma, float* dbeta, void* workspace,
                                           size_t workspace_bytes, int M, int D, int dtype, const float* delta_scale,
                                           int rows_per_scale, as_stream_t stream) {
  AS_REQUIRE(x && (dy || dx_res) && (dx_out || ddelta_out) && workspace, AS_E_BADARG, "as_add_layernorm_bwd: null pointer");
  AS_REQUIRE(rows_per_scale > 0, AS_E_BADARG, "as_add_layernorm_bwd: rows_per_scale must be positive");
  AS_REQUIRE(M > 0 && D > 0 && D % 4 == 0, AS_E_BADARG, "as_add_layernorm_bwd: need M > 0 and D %% 4 == 0 (D=%d)", D);
  AS_REQUIRE(workspace_bytes >= as_add_layernorm_bwd_workspace_bytes(M, D), AS_E_WORKSPACE,
             "as_add_layernorm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16)
    return launch_add_ln_bwd<__bf16>(x, dy, dx_res, gamma, dx_out, ddelta_out, dgamma, dbeta, (float*)workspace, M, D, eps,
                                     delta_scale, rows_per_scale, s);
  if (dtype == AS_F32)
    return launch_add_ln_bwd<float>(x, dy, dx_res, gamma, dx_out, ddelta_out, dgamma, dbeta, (float*)workspace, M, D, eps,
                                    delta_scale, rows_per_scale, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_add_layernorm_bwd: dtype %d", dtype);
}

extern "C" int as_add_layernorm(const float* x_in, const void* delta, const float* gamma, const float* beta, float eps,
                                float* x_out, void* y_out, int M, int D, int dtype, as_stream_t stream) {
  return as_add_layernorm_scaled(x_in, delta, gamma, beta, eps, x_out, y_out, M, D, dtype, nullptr, 1, stream);
}

extern "C" int as_add_layernorm_scaled(const float* x_in, const void* delta, const float* gamma, const float* beta, float eps,
                                       float* x_out, void* y_out, int M, int D, int dtype, const float* delta_scale,
                                       int rows_per_scale, as_stream_t stream) {
  AS_REQUIRE(x_in && (x_out || y_out), AS_E_BADARG, "as_add_layernorm: null pointer");
  AS_REQUIRE(rows_per_scale > 0 && (!delta_scale || delta), AS_E_BADARG, "as_add_layernorm: delta_scale needs delta, rows_per_scale > 0");
  AS_REQUIRE(!y_out || (gamma && beta), AS_E_BADARG, "as_add_layernorm: y_out needs gamma and beta");
  AS_REQUIRE(M > 0 && D > 0 && D % 4 == 0, AS_E_BADARG, "as_add_layernorm: need M > 0 and D %% 4 == 0 (D=%d)", D);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16) return launch_add_ln<__bf16>(x_in, delta, gamma, beta, x_out, y_out, M, D, eps, delta_scale, rows_per_scale, s);
  if (dtype == AS_F32) return launch_add_ln<float>(x_in, delta, gamma, beta, x_out, y_out, M, D, eps, delta_scale, rows_per_scale, s);
  AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_add_layernorm: dtype %d", dtype);
}


// ---------------------------------------------------------------------------------------------------------
// k x k / stride-k max pooling of a token-major (NHWC) fp32 feature map: the FPN's stride-32 tap
// (nn.MaxPool2d(2, 2) at patch 16, MaxPool2d(4, 4) / (2, 2) at patch 8; visual_transformer_det.py:107-127).  The taps live
// token-major here, so a thread owns 4 consecutive channels of one output pixel: k*k float4 loads, one store
// (ATen's NHWC kernel takes 42 us for the 25 MB of the ViT-B / 1024^2 tap; this one is a 31 MB stream).
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H,
                                                           int W, int C, int k, size_t xbs) {
  const int Ho = H / k, Wo = W / k, C4 = C >> 2;
  const size_t total = (size_t)B * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int ow = (int)(r % Wo);
    r /= Wo;
    const int oh = (int)(r % Ho), b = (int)(r / Ho);
    const float* src = x + (size_t)b * xbs + (((size_t)oh * k) * W + (size_t)ow * k) * C + c4 * 4;
    float4 m = *reinterpret_cast<const float4*>(src);
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)dy * W + dx) * C);
        // NaN propagates as in ATen's max_pool2d (a NaN in the window is the result)
        m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x; m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
        m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z; m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
      }
    *reinterpret_cast<float4*>(out + i * 4) = m;
  }
}
}  // namespace

extern "C" int as_maxpool_nhwc(const float* x, float* out, int B, int H, int W, int C, int k, long long x_batch_stride,
                               as_stream_t stream) {
  AS_REQUIRE(x && out, AS_E_BADARG, "as_maxpool_nhwc: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && k > 0, AS_E_BADARG, "as_maxpool_nhwc: bad sizes");
  AS_REQUIRE(x_batch_stride >= (long long)H * W * C && x_batch_stride % 4 == 0, AS_E_BADARG,
             "as_maxpool_nhwc: batch stride %lld (elements) must be >= H*W*C and a multiple of 4", x_batch_stride);
  AS_REQUIRE(C % 4 == 0 && H % k == 0 && W % k == 0, AS_E_UNSUPPORTED,
             "as_maxpool_nhwc: C %% 4 == 0 and H, W multiples of k only (C=%d H=%d W=%d k=%d)", C, H, W, k);
  const size_t total = (size_t)B * (H / k) * (W / k) * (C / 4);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, B, H, W, C, k,
                     (size_t)x_batch_stride);
  AS_CHECK_LAUNCH("maxpool_nhwc");
  return AS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Token assembly of prepare_tokens (visual_transformer_det.py:192-214): out[b, n] = table[n] (+ emb[b, n - 1] for the Np patch
// rows 1..Np), table = [cls + pos_0 ; pos_1..Np ; point tokens + their position embedding] kept by the caller.  One pass over
// the fp32 token tensor instead of an add and four slice writes.
// ---------------------------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const T* __restrict__ emb, const float* __restrict__ table,
                                                              float* __restrict__ out, int B, int Np, int N, int D) {
  const int D4 = D >> 2;
  const size_t total = (size_t)B * N * D4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D4) * 4;
    const size_t r = i / D4;
    const int n = (int)(r % N), b = (int)(r / N);
    float4 v = *reinterpret_cast<const float4*>(table + (size_t)n * D + c);
    if (n >= 1 && n <= Np) {
      const T* e = emb + ((size_t)b * Np + (n - 1)) * D + c;
      if (sizeof(T) == 2) {
        const bf16x4 e4 = *reinterpret_cast<const bf16x4*>(e);
        v.x = (float)e4[0] + v.x; v.y = (float)e4[1] + v.y; v.z = (float)e4[2] + v.z; v.w = (float)e4[3] + v.w;
      } else {
        const float4 e4 = *reinterpret_cast<const float4*>(e);
        v.x = e4.x + v.x; v.y = e4.y + v.y; v.z = e4.z + v.z; v.w = e4.w + v.w;
      }
    }
    *reinterpret_cast<float4*>(out + i * 4) = v;
  }
}
}  // namespace

extern "C" int as_assemble_tokens(const void* emb, const float* table, float* out, int B, int Np, int N, int D, int dtype,
                                  as_stream_t stream) {
  AS_REQUIRE(emb && table && out, AS_E_BADARG, "as_assemble_tokens: null pointer");
  AS_REQUIRE(B > 0 && Np > 0 && N > Np && D > 0 && D % 4 == 0, AS_E_BADARG, "as_assemble_tokens: B=%d Np=%d N=%d D=%d", B, Np, N, D);
  const size_t total = (size_t)B * N * (D / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == AS_BF16)
    hipLaunchKernelGGL(assemble_tokens_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)emb, table, out,
                       B, Np, N, D);
  else if (dtype == AS_F32)
    hipLaunchKernelGGL(assemble_tokens_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)emb, table, out,
                       B, Np, N, D);
  else
    AS_REQUIRE(false, AS_E_UNSUPPORTED, "as_assemble_tokens: dtype %d", dtype);
  AS_CHECK_LAUNCH("assemble_tokens");
  return AS_OK;
}
