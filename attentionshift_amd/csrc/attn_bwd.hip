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
  // fp32 partial products of the split-K weight-gradient GEMMs (bf16 only; the larger of the two: dWqkv)
  L.splitk_bytes = dtype == AS_BF16 ? as_linear_splitk_workspace_bytes(3 * D, D, (int)L.Mpad) : 0;
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
  STEP(transpose_pad<T>(dout, ws + L.off_doutT, M, D, Mpad, s));
  STEP(transpose_pad<T>(o, ws + L.off_oT, M, D, Mpad, s));
  // weight gradients contract over the TOKENS (K = Mpad = 8448 at config 2) into a few dozen output tiles: split-K with
  // fixed-order fp32 partials fills the chip (dWproj: 36 tiles -> 16 K ranges; dWqkv: 108 -> 7)
  if (sizeof(T) == 2)
    STEP(as_linear_splitk_fwd(ws + L.off_doutT, ws + L.off_oT, dWproj, D, D, Mpad, dtype, 0, ws + L.off_splitk, L.splitk_bytes, s));
  else
    STEP(as_linear_fwd(ws + L.off_doutT, ws + L.off_oT, nullptr, dWproj, D, D, Mpad, dtype, 0, s));   // dout^T . o
  if (dbproj) STEP(colsum<T>(dout, dbproj, (float*)(ws + L.off_part), M, D, s));
  // attention core
  STEP(as_sdpa_bwd(q, k, vt, o, d_o, lse, dqkv, ws + L.off_sdpa, as_sdpa_bwd_workspace_bytes(B, N, h, dtype), B, N, h,
                   dtype, s));
  // qkv backward
  STEP(transpose_pad<T>(Wqkv, ws + L.off_WqkvT, 3 * D, D, 3 * D, s));                         // [3D,D] -> [D,3D]
  STEP(as_linear_fwd(dqkv, ws + L.off_WqkvT, nullptr, dx, M, D, 3 * D, dtype, 0, s));         // dx = dqkv . Wqkv
  STEP(transpose_pad<T>(dqkv, ws + L.off_dqkvT, M, 3 * D, Mpad, s));
  STEP(transpose_pad<T>(x, ws + L.off_xT, M, D, Mpad, s));
  if (sizeof(T) == 2)
    STEP(as_linear_splitk_fwd(ws + L.off_dqkvT, ws + L.off_xT, dWqkv, 3 * D, D, Mpad, dtype, 0, ws + L.off_splitk, L.splitk_bytes, s));
  else
    STEP(as_linear_fwd(ws + L.off_dqkvT, ws + L.off_xT, nullptr, dWqkv, 3 * D, D, Mpad, dtype, 0, s)); // dqkv^T . x
  if (dbqkv) STEP(colsum<T>(dqkv, dbqkv, (float*)(ws + L.off_part), M, 3 * D, s));
#undef STEP
  return AS_OK;
}

}  // namespace

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
