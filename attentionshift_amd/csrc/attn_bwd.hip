// Backward of Attention.forward = qkv -> sdpa -> proj (reference models/vision_transformer.py:74-86) as one C call.
//
//   d_o     = dout . Wproj                      dWproj = dout^T . o          dbproj = colsum(dout)
//   dqkv    = as_sdpa_bwd(...)                  (sdpa_bwd.hip)
//   dx      = dqkv . Wqkv                       dWqkv  = dqkv^T . x          dbqkv  = colsum(dqkv)
//
// Every product runs on the forward GEMM kernel (gemm.hip: y = a . w^T, both operands contraction-contiguous), so the
// operands that are contracted over their ROW index are first transposed into the workspace (zero-padded to a
// multiple of 64 rows, the GEMM's K granularity): four activation transposes and two weight transposes per layer,
// about 0.1 GB of HBM traffic against 0.18 TFLOP of GEMM work at ViT-B / 1024^2 / B=2.
#include <algorithm>

#include "common.h"

namespace {

// in [R, C] row-major -> out [C, Rpad] row-major, columns R..Rpad-1 zero
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int C,
                                                            int Rpad) {
  __shared__ T tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : from_f32<T>(0.0f);
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) out[(size_t)c * Rpad + r] = tile[tx][i];
  }
}

// The same for bf16 with C % 8 == 0, moving 16 bytes per lane on both sides: the tile is staged in LDS as packed pairs
// [r][c/2] (pitch 33 words); a lane of the store phase gathers the 8 rows of its output vector from one word column
// (8 row groups x 4 word columns per wave: 32 banks, the two halves of a word share a broadcast).
__global__ __launch_bounds__(256) void transpose_pad_vec_kernel(const __bf16* __restrict__ in, __bf16* __restrict__ out, int R,
                                                                int C, int Rpad) {
  __shared__ uint32_t tile[64][33];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = t + 256 * i, row = v >> 3, cv = v & 7;
    const int r = r0 + row, c = c0 + cv * 8;
    uint4 d = make_uint4(0u, 0u, 0u, 0u);
    if (r < R && c < C) d = *reinterpret_cast<const uint4*>(in + (size_t)r * C + c);
    tile[row][cv * 4 + 0] = d.x;
    tile[row][cv * 4 + 1] = d.y;
    tile[row][cv * 4 + 2] = d.z;
    tile[row][cv * 4 + 3] = d.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = t + 256 * i, cl = v >> 3, ch = v & 7;
    const int c = c0 + cl;
    if (c >= C) continue;
    const int w = cl >> 1, sh = (cl & 1) * 16;
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (tile[ch * 8 + j][w] >> sh) & 0xffffu;
    const uint4 o = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    *reinterpret_cast<uint4*>(out + (size_t)c * Rpad + r0 + ch * 8) = o;
  }
}

// out[r] = fp32 sum of row r of a bf16 matrix [R, P] (P % 8 == 0): the bias gradient read from the TRANSPOSED, zero-padded
// upstream gradient the weight-gradient product needs anyway -- contiguous 16-byte loads, one workgroup per row, a fixed
// summation order (lane-strided partials, then a tree): bit-reproducible.
__global__ __launch_bounds__(256) void rowsum_bf16_kernel(const __bf16* __restrict__ in, float* __restrict__ out, int P) {
  __shared__ float sm[256];
  const __bf16* row = in + (size_t)blockIdx.x * P;
  float s = 0.0f;
  for (int i = threadIdx.x * 8; i < P; i += 256 * 8) {
    const uint4 d = *reinterpret_cast<const uint4*>(row + i);
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) s += __uint_as_float(w[j] << 16) + __uint_as_float(w[j] & 0xffff0000u);
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

// fp32 column sums of in [R, C] in two fixed-order stages: CS_SLICES row slices -> partials [CS_SLICES, C] (in the
// workspace), then one pass over the partials.  No atomics: bit-reproducible.
constexpr int CS_SLICES = 64;
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ in, float* __restrict__ part, int R,
                                                             int C) {
  __shared__ float sm[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  const int rows = as_ceil_div_dev(R, CS_SLICES);
  const int r0 = blockIdx.y * rows, r1 = min(R, r0 + rows);
  float s = 0.0f;
  if (c < C)
    for (int r = r0 + ty; r < r1; r += 4) s += to_f32<T>(in[(size_t)r * C + c]);
  sm[ty][threadIdx.x & 63] = s;
  __syncthreads();
  if (ty == 0 && c < C)
    part[(size_t)blockIdx.y * C + c] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.0f;
  for (int i = 0; i < CS_SLICES; ++i) s += part[(size_t)i * C + c];
  out[c] = s;
}

template <typename T> int transpose_pad(const void* in, void* out, int R, int C, int Rpad, hipStream_t s) {
  dim3 grid(as_ceil_div(Rpad, 64), as_ceil_div(C, 64));
  if constexpr (sizeof(T) == 2) {
    if (C % 8 == 0 && Rpad % 64 == 0) {
      hipLaunchKernelGGL(transpose_pad_vec_kernel, grid, dim3(256), 0, s, (const __bf16*)in, (__bf16*)out, R, C, Rpad);
      AS_CHECK_LAUNCH("transpose_pad_vec");
      return AS_OK;
    }
  }
  hipLaunchKernelGGL((transpose_pad_kernel<T>), grid, dim3(256), 0, s, (const T*)in, (T*)out, R, C, Rpad);
  AS_CHECK_LAUNCH("transpose_pad");
  return AS_OK;
}
template <typename T> int colsum(const void* in, float* out, float* part, int R, int C, hipStream_t s) {
  hipLaunchKernelGGL((colsum_partial_kernel<T>), dim3(as_ceil_div(C, 64), CS_SLICES), dim3(256), 0, s, (const T*)in, part,
                     R, C);
  AS_CHECK_LAUNCH("colsum_partial");
  hipLaunchKernelGGL(colsum_final_kernel, dim3(as_ceil_div(C, 256)), dim3(256), 0, s, (const float*)part, out, C);
  AS_CHECK_LAUNCH("colsum_final");
  return AS_OK;
}

// db from the transposed gradient [C, Rpad] (bf16) when the caller has it, else the two-stage column sums
template <typename T> int bias_grad(const void* g, const void* gT, float* out, float* part, int R, int C, int Rpad, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    if (gT) {
      hipLaunchKernelGGL(rowsum_bf16_kernel, dim3(C), dim3(256), 0, s, (const __bf16*)gT, out, Rpad);
      AS_CHECK_LAUNCH("rowsum_bf16");
      return AS_OK;
    }
  }
  return colsum<T>(g, out, part, R, C, s);
}

struct BwdLayout {
  size_t es, M, Mpad, D;
  size_t off_sdpa, off_do, off_dqkv, off_doutT, off_oT, off_dqkvT, off_xT, off_WprojT, off_WqkvT, off_part, off_splitk,
      splitk_bytes, total;
};
BwdLayout bwd_layout(int B, int N, int D, int h, int dtype) {
  BwdLayout L{};
  L.es = dtype == AS_F32 ? 4 : 2;
  L.M = (size_t)B * N;
  L.Mpad = as_round_up((int)L.M, 64);
  L.D = D;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  L.off_sdpa = take(as_sdpa_bwd_workspace_bytes(B, N, h, dtype));
  L.off_do = take(L.M * D * L.es);
  L.off_dqkv = take(L.M * 3 * D * L.es);
  L.off_doutT = take((size_t)D * L.Mpad * L.es);
  L.off_oT = take((size_t)D * L.Mpad * L.es);
  L.off_dqkvT = take((size_t)3 * D * L.Mpad * L.es);
  L.off_xT = take((size_t)D * L.Mpad * L.es);
  L.off_WprojT = take((size_t)D * D * L.es);
  L.off_WqkvT = take((size_t)3 * D * D * L.es);
  L.off_part = take((size_t)CS_SLICES * 3 * D * sizeof(float));
  // fp32 partial products of the split-K weight-gradient GEMMs (bf16 only; the larger of the two)
  L.splitk_bytes = dtype == AS_BF16 ? std::max(as_linear_splitk_workspace_bytes(3 * D, D, (int)L.Mpad),
                                               as_linear_splitk_workspace_bytes(D, D, (int)L.Mpad))
                                    : 0;
  if (dtype == AS_BF16 && as_tn_applies((int)L.M, 3 * D, D) && as_tn_applies((int)L.M, D, D))
    L.splitk_bytes = std::max(L.splitk_bytes, std::max(as_tn_workspace_bytes((int)L.M, 3 * D, D), as_tn_workspace_bytes((int)L.M, D, D)));
  L.off_splitk = take(L.splitk_bytes);
  L.total = off;
  return L;
}

template <typename T>
int attn_bwd(const void* x, const void* Wqkv, const void* Wproj, const void* dout, const void* q, const void* k,
             const void* vt, const void* o, const float* lse, void* dx, void* dWqkv, float* dbqkv, void* dWproj,
             float* dbproj, char* ws, const BwdLayout& L, int B, int N, int D, int h, int dtype, hipStream_t s) {
  const int M = (int)L.M, Mpad = (int)L.Mpad;
  void* d_o = ws + L.off_do;
  void* dqkv = ws + L.off_dqkv;
  int rc;
#define STEP(call)            \
  do {                        \
    rc = (call);              \
    if (rc != AS_OK) return rc; \
  } while (0)
  // proj backward
  STEP(transpose_pad<T>(Wproj, ws + L.off_WprojT, D, D, D, s));                               // [Dout,Din] -> [Din,Dout]
  STEP(as_linear_fwd(dout, ws + L.off_WprojT, nullptr, d_o, M, D, D, dtype, 0, s));           // d_o = dout . Wproj
  // weight gradients contract over the TOKENS (K = Mpad = 8448 at config 2) into a few dozen output tiles: split over the
  // token range with fixed-order fp32 partials (dWproj: 36 tiles -> 7 ranges; dWqkv: 108 -> 2).  bf16 with 128-aligned
  // widths: straight from the row-major activations (csrc/gemm_tn.hip); else through transposed, zero-padded copies
  const bool tn = sizeof(T) == 2 && as_tn_applies(M, D, D) && as_tn_applies(M, 3 * D, D) &&
                  L.splitk_bytes >= as_tn_workspace_bytes(M, 3 * D, D) && L.splitk_bytes >= as_tn_workspace_bytes(M, D, D);
  if (tn) {
    STEP(as_tn_dw(dout, o, dWproj, dbproj, (float*)(ws + L.off_part), M, D, D, 0, ws + L.off_splitk, L.splitk_bytes, s));
  } else {
    STEP(transpose_pad<T>(dout, ws + L.off_doutT, M, D, Mpad, s));
    STEP(transpose_pad<T>(o, ws + L.off_oT, M, D, Mpad, s));
    if (sizeof(T) == 2)
      STEP(as_linear_splitk_fwd(ws + L.off_doutT, ws + L.off_oT, dWproj, D, D, Mpad, dtype, 0, ws + L.off_splitk, L.splitk_bytes, s));
    else
      STEP(as_linear_fwd(ws + L.off_doutT, ws + L.off_oT, nullptr, dWproj, D, D, Mpad, dtype, 0, s));   // dout^T . o
    if (dbproj) STEP(bias_grad<T>(dout, ws + L.off_doutT, dbproj, (float*)(ws + L.off_part), M, D, Mpad, s));
  }
  // attention core
  STEP(as_sdpa_bwd(q, k, vt, o, d_o, lse, dqkv, ws + L.off_sdpa, as_sdpa_bwd_workspace_bytes(B, N, h, dtype), B, N, h,
                   dtype, s));
  // qkv backward
  STEP(transpose_pad<T>(Wqkv, ws + L.off_WqkvT, 3 * D, D, 3 * D, s));                         // [3D,D] -> [D,3D]
  STEP(as_linear_fwd(dqkv, ws + L.off_WqkvT, nullptr, dx, M, D, 3 * D, dtype, 0, s));         // dx = dqkv . Wqkv
  if (tn) {
    STEP(as_tn_dw(dqkv, x, dWqkv, dbqkv, (float*)(ws + L.off_part), M, 3 * D, D, 0, ws + L.off_splitk, L.splitk_bytes, s));
  } else {
    STEP(transpose_pad<T>(dqkv, ws + L.off_dqkvT, M, 3 * D, Mpad, s));
    STEP(transpose_pad<T>(x, ws + L.off_xT, M, D, Mpad, s));
    if (sizeof(T) == 2)
      STEP(as_linear_splitk_fwd(ws + L.off_dqkvT, ws + L.off_xT, dWqkv, 3 * D, D, Mpad, dtype, 0, ws + L.off_splitk, L.splitk_bytes, s));
    else
      STEP(as_linear_fwd(ws + L.off_dqkvT, ws + L.off_xT, nullptr, dWqkv, 3 * D, D, Mpad, dtype, 0, s)); // dqkv^T . x
    if (dbqkv) STEP(bias_grad<T>(dqkv, ws + L.off_dqkvT, dbqkv, (float*)(ws + L.off_part), M, 3 * D, Mpad, s));
  }
#undef STEP
  return AS_OK;
}

// Backward of one nn.Linear (y = x W^T + b) with the same three pieces: dx = dy . W on the forward kernel against a
// transposed copy of W, dW = dy^T . x as a split-K product of the transposed activations, db = colsum(dy).
struct LinBwdLayout {
  size_t Mpad, off_WT, off_dyT, off_xT, off_part, off_splitk, splitk_bytes, total;
};
LinBwdLayout lin_bwd_layout(int M, int Nout, int K) {
  LinBwdLayout L{};
  L.Mpad = as_round_up(M, 64);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  L.off_WT = take((size_t)K * Nout * 2);
  L.off_dyT = take((size_t)Nout * L.Mpad * 2);
  L.off_xT = take((size_t)K * L.Mpad * 2);
  L.off_part = take((size_t)CS_SLICES * Nout * sizeof(float));
  L.splitk_bytes = as_linear_splitk_workspace_bytes(Nout, K, (int)L.Mpad);
  if (as_tn_applies(M, Nout, K) && as_tn_workspace_bytes(M, Nout, K) > L.splitk_bytes) L.splitk_bytes = as_tn_workspace_bytes(M, Nout, K);
  L.off_splitk = take(L.splitk_bytes);
  L.total = off;
  return L;
}

}  // namespace

extern "C" size_t as_linear_bwd_workspace_bytes(int M, int Nout, int K) {
  if (M <= 0 || Nout <= 0 || K <= 0) return 0;
  return lin_bwd_layout(M, Nout, K).total;
}

extern "C" int as_linear_bwd(const void* x, const void* W, const void* dy, void* dx, void* dW, float* db, int M, int Nout,
                             int K, int dtype, int dw_f32, void* workspace, size_t workspace_bytes, as_stream_t stream) {
  return as_linear_bwd_dgelu(x, W, dy, nullptr, dx, dW, db, M, Nout, K, dtype, dw_f32, workspace, workspace_bytes, stream);
}

// as_linear_bwd whose input gradient is multiplied by gelu'(pre) in the GEMM's epilogue (pre [M,K] = the pre-activation the
// linear's input was the GELU of; NULL = plain as_linear_bwd): dx is then the gradient of the PRE-activation
extern "C" int as_linear_bwd_dgelu(const void* x, const void* W, const void* dy, const void* pre, void* dx, void* dW, float* db,
                                   int M, int Nout, int K, int dtype, int dw_f32, void* workspace, size_t workspace_bytes,
                                   as_stream_t stream) {
  AS_REQUIRE(dy && workspace && (dx || dW || db), AS_E_BADARG, "as_linear_bwd: null pointer");
  AS_REQUIRE(!pre || (dx && K % 8 == 0 && Nout % 32 == 0), AS_E_BADARG, "as_linear_bwd_dgelu: pre needs dx, K %% 8 == 0");
  AS_REQUIRE(!dx || W, AS_E_BADARG, "as_linear_bwd: dx needs W");
  AS_REQUIRE(!dW || x, AS_E_BADARG, "as_linear_bwd: dW needs x");
  AS_REQUIRE(dtype == AS_BF16, AS_E_UNSUPPORTED, "as_linear_bwd: bf16 operands only (dtype %d)", dtype);
  AS_REQUIRE(M > 0 && Nout > 0 && K > 0 && Nout % 32 == 0 && K % 4 == 0, AS_E_BADARG,
             "as_linear_bwd: need M > 0, Nout %% 32 == 0, K %% 4 == 0 (M=%d Nout=%d K=%d)", M, Nout, K);
  const LinBwdLayout L = lin_bwd_layout(M, Nout, K);
  AS_REQUIRE(workspace_bytes >= L.total, AS_E_BADARG, "as_linear_bwd: workspace too small (%zu < %zu)", workspace_bytes,
             L.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int Mpad = (int)L.Mpad;
  int rc;
  if (dx) {
    if ((rc = transpose_pad<__bf16>(W, ws + L.off_WT, Nout, K, Nout, s)) != AS_OK) return rc;        // [Nout,K] -> [K,Nout]
    if (pre) rc = as_linear_dgelu_fwd(dy, ws + L.off_WT, pre, dx, M, K, Nout, dtype, s);
    else rc = as_linear_fwd(dy, ws + L.off_WT, nullptr, dx, M, K, Nout, dtype, 0, s);
    if (rc != AS_OK) return rc;
  }
  // dW = dy^T . x straight from the row-major activations (csrc/gemm_tn.hip: transposing LDS reads) when the feature counts
  // are 128-aligned; otherwise (and with AS_BWD_TRANSPOSED=1) through transposed, zero-padded copies as in round 3
  const bool tn = as_tn_applies(M, Nout, K);
  if (dW && tn) {                       // (db rides in the same pass)
    if ((rc = as_tn_dw(dy, x, dW, db, (float*)(ws + L.off_part), M, Nout, K, dw_f32, ws + L.off_splitk, L.splitk_bytes, s)) != AS_OK) return rc;
  } else if (dW) {
    if ((rc = transpose_pad<__bf16>(dy, ws + L.off_dyT, M, Nout, Mpad, s)) != AS_OK) return rc;
    if ((rc = transpose_pad<__bf16>(x, ws + L.off_xT, M, K, Mpad, s)) != AS_OK) return rc;
    if ((rc = as_linear_splitk_fwd(ws + L.off_dyT, ws + L.off_xT, dW, Nout, K, Mpad, dtype, dw_f32, ws + L.off_splitk,
                                   L.splitk_bytes, s)) != AS_OK)
      return rc;
  }
  if (db && tn && dW) {
    // (done above)
  } else if (db && tn) {
    if ((rc = as_tn_colsum(dy, db, (float*)(ws + L.off_part), M, Nout, s)) != AS_OK) return rc;
  } else if (db && (rc = bias_grad<__bf16>(dy, dW ? ws + L.off_dyT : nullptr, db, (float*)(ws + L.off_part), M, Nout, Mpad, s)) != AS_OK) {
    return rc;
  }
  return AS_OK;
}

extern "C" size_t as_attn_bwd_workspace_bytes(int B, int N, int D, int h, int dtype) {
  if (B <= 0 || N <= 0 || D <= 0 || h <= 0) return 0;
  return bwd_layout(B, N, D, h, dtype).total;
}

extern "C" int as_attn_bwd(const void* x, const void* Wqkv, const void* Wproj, const void* dout, const void* q,
                           const void* k, const void* vt, const void* o, const float* lse, void* dx, void* dWqkv,
                           float* dbqkv, void* dWproj, float* dbproj, void* workspace, size_t workspace_bytes, int B,
                           int N, int D, int h, int dtype, as_stream_t stream) {
  AS_REQUIRE(x && Wqkv && Wproj && dout && q && k && vt && o && lse && dx && dWqkv && dWproj && workspace, AS_E_BADARG,
             "as_attn_bwd: null pointer");
  AS_REQUIRE(B > 0 && N > 0 && h > 0 && D == h * AS_HEAD_DIM, AS_E_UNSUPPORTED,
             "as_attn_bwd: head dim must be 64 (D=%d h=%d)", D, h);
  AS_REQUIRE(dtype == AS_F32 || dtype == AS_BF16, AS_E_UNSUPPORTED, "as_attn_bwd: dtype %d", dtype);
  const BwdLayout L = bwd_layout(B, N, D, h, dtype);
  AS_REQUIRE(workspace_bytes >= L.total, AS_E_BADARG, "as_attn_bwd: workspace too small (%zu < %zu)", workspace_bytes,
             L.total);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AS_BF16)
    return attn_bwd<__bf16>(x, Wqkv, Wproj, dout, q, k, vt, o, lse, dx, dWqkv, dbqkv, dWproj, dbproj, (char*)workspace, L,
                            B, N, D, h, dtype, s);
  return attn_bwd<float>(x, Wqkv, Wproj, dout, q, k, vt, o, lse, dx, dWqkv, dbqkv, dWproj, dbproj, (char*)workspace, L, B,
                         N, D, h, dtype, s);
}
