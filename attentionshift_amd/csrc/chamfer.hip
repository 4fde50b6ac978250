// 2-D chamfer distance for gfx950: the reference's second native op (mmdet/ops/chamfer_2d/src/chamfer_2d.cu:12-161,
// bound by mmdet/ops/chamfer_2d/dist_chamfer_2d.py).  Off the hot path (only the RepPoints variants call it) and the
// reference does not build it, so parity is by definition: for every point the SQUARED distance to its nearest
// neighbour in the other set and that neighbour's index (lowest index on ties, as the reference's strict `<`), and
// the gradient 2 * g * (p - q) of both points -- GATHERED per point in a fixed order (round 4: the float atomics of the
// first version were the library's only non-deterministic reduction).  (The upstream kernel reads up to two stale shared-memory
// entries when the neighbour count is 2 or 3 mod 4 -- `end_k & 2` where `end_k % 4` was meant; that is not
// reproduced.)
#include "common.h"

namespace {

constexpr int CH_NT = 256;
constexpr int CH_TILE = 1024;      // neighbour points staged in LDS per pass

// grid (ceil(n / CH_NT), B): thread = one query point of set 1, neighbours of set 2 streamed through LDS
__global__ __launch_bounds__(CH_NT) void chamfer_nn_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                           float* __restrict__ dist, int32_t* __restrict__ idx, int n,
                                                           int m) {
  __shared__ float2 buf[CH_TILE];
  const int b = blockIdx.y, j = blockIdx.x * CH_NT + threadIdx.x;
  const float2* q = reinterpret_cast<const float2*>(xyz2) + (size_t)b * m;
  float2 p = make_float2(0.0f, 0.0f);
  if (j < n) p = reinterpret_cast<const float2*>(xyz1)[(size_t)b * n + j];
  float best = INFINITY;
  int best_i = 0;
  for (int k0 = 0; k0 < m; k0 += CH_TILE) {
    const int cnt = min(CH_TILE, m - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += CH_NT) buf[k] = q[k0 + k];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const float dx = buf[k].x - p.x, dy = buf[k].y - p.y;
      const float d = dx * dx + dy * dy;
      if (d < best) { best = d; best_i = k0 + k; }           // strict: ties keep the lowest index
    }
  }
  if (j < n) { dist[(size_t)b * n + j] = best; idx[(size_t)b * n + j] = best_i; }
}

// grid (ceil(nA / CH_NT), B): the WHOLE gradient of one point set, gathered -- no atomics, a fixed summation order.
// Point j of set A receives 2 g_A[j] (a_j - b_idxA[j]) from its own nearest-neighbour term and, from every point k of
// set B whose nearest neighbour it is (idxB[k] == j), -2 g_B[k] (b_k - a_j); the B side streams through LDS in
// tiles and is scanned in index order by every thread (O(nA nB), the same work as the forward's search).
__global__ __launch_bounds__(CH_NT) void chamfer_grad_gather_kernel(const float* __restrict__ xyzA, const float* __restrict__ xyzB,
                                                                    const float* __restrict__ gdistA, const int32_t* __restrict__ idxA,
                                                                    const float* __restrict__ gdistB, const int32_t* __restrict__ idxB,
                                                                    float* __restrict__ gA, int nA, int nB) {
  __shared__ float2 pts[CH_TILE];
  __shared__ float gb[CH_TILE];
  __shared__ int ib[CH_TILE];
  const int b = blockIdx.y, j = blockIdx.x * CH_NT + threadIdx.x;
  const float2* A = reinterpret_cast<const float2*>(xyzA) + (size_t)b * nA;
  const float2* Bp = reinterpret_cast<const float2*>(xyzB) + (size_t)b * nB;
  float2 a = make_float2(0.0f, 0.0f);
  float ax = 0.0f, ay = 0.0f;
  if (j < nA) {
    a = A[j];
    const float2 q = Bp[idxA[(size_t)b * nA + j]];
    const float g = gdistA[(size_t)b * nA + j] * 2.0f;
    ax = g * (a.x - q.x);
    ay = g * (a.y - q.y);
  }
  for (int k0 = 0; k0 < nB; k0 += CH_TILE) {
    const int cnt = min(CH_TILE, nB - k0);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += CH_NT) {
      pts[k] = Bp[k0 + k];
      gb[k] = gdistB[(size_t)b * nB + k0 + k];
      ib[k] = idxB[(size_t)b * nB + k0 + k];
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k)
      if (ib[k] == j) {
        const float g = gb[k] * 2.0f;
        ax += g * (a.x - pts[k].x);
        ay += g * (a.y - pts[k].y);
      }
  }
  if (j < nA) reinterpret_cast<float2*>(gA)[(size_t)b * nA + j] = make_float2(ax, ay);
}

}  // namespace

extern "C" int as_chamfer_2d_fwd(const float* xyz1, const float* xyz2, float* dist1, float* dist2, int32_t* idx1,
                                 int32_t* idx2, int B, int n, int m, as_stream_t stream) {
  AS_REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2, AS_E_BADARG, "as_chamfer_2d_fwd: null pointer");
  AS_REQUIRE(B > 0 && n > 0 && m > 0, AS_E_BADARG, "as_chamfer_2d_fwd: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(chamfer_nn_kernel, dim3(as_ceil_div(n, CH_NT), B), dim3(CH_NT), 0, s, xyz1, xyz2, dist1, idx1, n, m);
  hipLaunchKernelGGL(chamfer_nn_kernel, dim3(as_ceil_div(m, CH_NT), B), dim3(CH_NT), 0, s, xyz2, xyz1, dist2, idx2, m, n);
  AS_CHECK_LAUNCH("chamfer_2d_fwd");
  return AS_OK;
}

extern "C" int as_chamfer_2d_bwd(const float* xyz1, const float* xyz2, const float* gdist1, const float* gdist2,
                                 const int32_t* idx1, const int32_t* idx2, float* gxyz1, float* gxyz2, int B, int n, int m,
                                 as_stream_t stream) {
  AS_REQUIRE(xyz1 && xyz2 && gdist1 && gdist2 && idx1 && idx2 && gxyz1 && gxyz2, AS_E_BADARG,
             "as_chamfer_2d_bwd: null pointer");
  AS_REQUIRE(B > 0 && n > 0 && m > 0, AS_E_BADARG, "as_chamfer_2d_bwd: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(chamfer_grad_gather_kernel, dim3(as_ceil_div(n, CH_NT), B), dim3(CH_NT), 0, s, xyz1, xyz2, gdist1, idx1, gdist2,
                     idx2, gxyz1, n, m);
  hipLaunchKernelGGL(chamfer_grad_gather_kernel, dim3(as_ceil_div(m, CH_NT), B), dim3(CH_NT), 0, s, xyz2, xyz1, gdist2, idx2, gdist1,
                     idx1, gxyz2, m, n);
  AS_CHECK_LAUNCH("chamfer_2d_bwd");
  return AS_OK;
}
