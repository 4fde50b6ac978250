// Final similarity of the mean-shift stage over the UNMASKED patch grid (reference stdroi:848-851, the last
// F.cosine_similarity of cosine_shift_batch; clamp(0) is the caller's, :1840).
#include "common.h"

namespace {

constexpr int SC_NT = 512;
constexpr int SC_NW = SC_NT / 64;
constexpr int SC_PMAX = 32;
constexpr float SC_EPS = 1e-8f;

struct ScBox { int x0, y0, x1, y1; };
__device__ __forceinline__ ScBox sc_box(const int32_t* bp, int g, int Hp, int Wp) {
  ScBox b;
  b.x0 = max(bp[g * 4 + 0], 0); b.y0 = max(bp[g * 4 + 1], 0);
  b.x1 = min(bp[g * 4 + 2], Wp - 1); b.y1 = min(bp[g * 4 + 3], Hp - 1);
  return b;
}
__device__ __forceinline__ int sc_bw(const ScBox& b) { return max(b.x1 - b.x0 + 1, 0); }
__device__ __forceinline__ bool sc_inside(const ScBox& b, int n, int Wp) {
  const int y = n / Wp, x = n - y * Wp;
  return x >= b.x0 && x <= b.x1 && y >= b.y0 && y <= b.y1;
}

// ---- final similarity over the UNMASKED grid: grid (tiles of 32 patches, B), 512 threads -----------------------------
// The prototypes of all objects of the image are packed into 32-row tiles (3 objects x 20 prototypes = 60 rows = 2
// tiles, not 3); the tile's feature fragments are loaded once, both operands' norms are accumulated from the MFMA
// fragments (no norm pass over the map), the 8 waves split the channel range.  With aw != null it also reduces the
// density sums of the last assignment per (tile, object, prototype) for the tau trace.
constexpr int SF2_SU = 8;             // k16 steps per wave held in registers: C <= 8 * 8 * 16 = 1024

__global__ __launch_bounds__(SC_NT) void shift_final_sim_kernel(const float* __restrict__ feat, const float* __restrict__ prot,
                                                                const int32_t* __restrict__ box_patch,
                                                                const int32_t* __restrict__ obj_img,
                                                                const int2* __restrict__ aw, float* __restrict__ sim,
                                                                float* __restrict__ part_stats, int C, int Hp, int Wp,
                                                                int P, int G, int nt1, size_t fbs) {
  __shared__ float red[SC_NW][32][33];
  __shared__ float nrmA[2][2 * SC_NW][32], nrmB[2 * SC_NW][32];
  __shared__ int objs[256];
  __shared__ int nobj_s;
  const int Np = Hp * Wp;
  const int b = blockIdx.y, tile = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int n_mine = min(tile * 32 + li, Np - 1);
  const float* frow = feat + (size_t)b * fbs + (size_t)n_mine * C;
  const int nsteps = C / 16;

  float fb[SF2_SU][8];
  bool okk[SF2_SU];
#pragma unroll
  for (int u = 0; u < SF2_SU; ++u) {
    const int su = wave + SC_NW * u;
    okk[u] = su < nsteps;
    const float* p = frow + (okk[u] ? su * 16 + half * 8 : 0);
    const float4 a = *reinterpret_cast<const float4*>(p), c4 = *reinterpret_cast<const float4*>(p + 4);
    fb[u][0] = a.x; fb[u][1] = a.y; fb[u][2] = a.z; fb[u][3] = a.w; fb[u][4] = c4.x; fb[u][5] = c4.y; fb[u][6] = c4.z; fb[u][7] = c4.w;
  }
  if (tid == 0) {
    int n = 0;
    for (int g = 0; g < G && n < 256; ++g)
      if (obj_img[g] == b) objs[n++] = g;
    nobj_s = n;
  }
  __builtin_amdgcn_sched_barrier(0);
  float qb = 0.0f;
#pragma unroll
  for (int u = 0; u < SF2_SU; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      fb[u][j] = okk[u] ? fb[u][j] : 0.0f;
      qb = fmaf(fb[u][j], fb[u][j], qb);
    }
  nrmB[wave * 2 + half][li] = qb;
  __syncthreads();
  const int rows = nobj_s * P;
  const int nn = tid & 31, pq = tid >> 5;                      // epilogue: thread = (patch, 16 rows per pass)
  const int n_out = tile * 32 + nn;
  float fin;
  {
    float s = 0.0f;
#pragma unroll
    for (int kq = 0; kq < 2 * SC_NW; ++kq) s += nrmB[kq][nn];
    fin = 1.0f / fmaxf(sqrtf(s), SC_EPS);
  }

  for (int rt0 = 0; rt0 * 32 < rows; rt0 += 2) {
    f32x16 acc[2];
    const bool two = (rt0 + 1) * 32 < rows;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int r = (rt0 + tt) * 32 + li;
      const bool rv = r < rows && (tt == 0 || two);
      const int rr = rv ? r : 0;
      const int gi = objs[min(rr / P, 255)], p = rr - (rr / P) * P;
      const float* prow = prot + ((size_t)gi * P + p) * C;
      float fa[SF2_SU][8];
#pragma unroll
      for (int u = 0; u < SF2_SU; ++u) {
        const float* ap = prow + (okk[u] ? (wave + SC_NW * u) * 16 + half * 8 : 0);
        const float4 a = *reinterpret_cast<const float4*>(ap), c4 = *reinterpret_cast<const float4*>(ap + 4);
        fa[u][0] = a.x; fa[u][1] = a.y; fa[u][2] = a.z; fa[u][3] = a.w; fa[u][4] = c4.x; fa[u][5] = c4.y; fa[u][6] = c4.z; fa[u][7] = c4.w;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r2 = 0; r2 < 16; ++r2) acc[tt][r2] = 0.0f;
      float qa = 0.0f;
      if (tt == 0 || two) {
#pragma unroll
        for (int u = 0; u < SF2_SU; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float av = (okk[u] && rv) ? fa[u][j] : 0.0f;
            qa = fmaf(av, av, qa);
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, fb[u][j], acc[tt], 0, 0, 0);      // D[row][n]
          }
      }
      nrmA[tt][wave * 2 + half][li] = qa;
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      if (tt == 1 && !two) break;                               // workgroup-uniform
#pragma unroll
      for (int r2 = 0; r2 < 16; ++r2) red[wave][acc_row(r2, half)][li] = acc[tt][r2];
      __syncthreads();
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        const int rl = pq + 16 * qd;
        const int r = (rt0 + tt) * 32 + rl;
        const bool rv = r < rows;                               // uniform over the 32 lanes that share rl
        float v = 0.0f;
        int gi = 0, p = 0;
        if (rv) {
          gi = objs[r / P]; p = r - (r / P) * P;
          float sn = 0.0f;
#pragma unroll
          for (int kq = 0; kq < 2 * SC_NW; ++kq) sn += nrmA[tt][kq][rl];
          v = (((red[0][rl][nn] + red[1][rl][nn]) + (red[2][rl][nn] + red[3][rl][nn])) +
               ((red[4][rl][nn] + red[5][rl][nn]) + (red[6][rl][nn] + red[7][rl][nn])));
          v = v * (1.0f / fmaxf(sqrtf(sn), SC_EPS)) * fin;
          if (n_out < Np) sim[((size_t)gi * P + p) * Np + n_out] = v;
        }
        if (aw != nullptr) {                                    // density sums of the last assignment (tau trace)
          float ds = 0.0f;
          if (rv && n_out < Np) {
            const ScBox ob = sc_box(box_patch, gi, Hp, Wp);
            if (sc_inside(ob, n_out, Wp)) {
              const int y = n_out / Wp, x = n_out - y * Wp;
              if (aw[(size_t)gi * Np + (y - ob.y0) * sc_bw(ob) + (x - ob.x0)].x == p) ds = v;
            }
          }
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) ds += __shfl_xor(ds, o);
          if (rv && nn == 0) {
            float* ps = part_stats + (((size_t)gi * nt1 + tile) * SC_PMAX + p) * 2;
            ps[0] = 0.0f;
            ps[1] = ds;
          }
        }
      }
      __syncthreads();                                          // red / nrmA are rewritten by the next tile
    }
  }
}

}  // namespace

// grid (tiles of 32 patches, B).  aw != null: also the density sums of the last assignment (tau trace).
void as_shift_final_sim_launch(const float* feat, const float* prot, const int32_t* box_patch, const int32_t* obj_img,
                               const int2* aw, float* sim_out, float* part_stats, int B, int C, int Hp, int Wp, int P, int G,
                               int nt1, size_t fbs, hipStream_t s) {
  hipLaunchKernelGGL(shift_final_sim_kernel, dim3(as_ceil_div(Hp * Wp, 32), B), dim3(SC_NT), 0, s, feat, prot, box_patch,
                     obj_img, aw, sim_out, part_stats, C, Hp, Wp, P, G, nt1, fbs);
}
