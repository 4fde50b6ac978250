// Mean-shift token clustering ("attention shift" iteration) for gfx950, fp32 throughout.
//
// Restates cosine_shift_batch + update_density_batch
// (reference mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py:830-854, 882-908) with the
// box masking of mean_shift_grid_prototype (:1819-1824) folded in: features outside an object's box are
// zero in the reference, so their cosine is exactly 0, they add exp(-max) to every softmax denominator,
// nothing to the aggregation, and all fall into ONE cluster (the argmax of exp(-max_p)/Z_p).  The loop
// therefore only touches in-box patches; the final similarity map is over the whole (unmasked) grid.
//
// Per iteration (deterministic, no float atomics; every cross-workgroup reduction goes through
// partials summed in a fixed order):
//   sim      : cos(prot, feat) tiles, 32 patches x <=32 prototypes, exact-fp32 MFMA (v_mfma_f32_32x32x2),
//              the 4 waves of a workgroup split the channel range; also per-tile max and the density sums
//              of the PREVIOUS assignment (update_density_batch needs cos(new prot, feat) = this pass)
//   stats    : tau, logit max, softmax denominator Z per prototype
//   assign   : w = exp(sim/(temp*tau) - max)/Z, argmax over prototypes (ties -> lowest), one-hot
//              weighted aggregation prot_new[a] += w * feat[n] with register accumulators
//   finalize : sum the per-tile partial prototypes, norms, member counts
#include "common.h"

namespace {

constexpr int CS_NT = 256;
constexpr int CS_TILE1 = 32;     // patches per sim tile
constexpr int CS_TILE3 = 32;     // patches per aggregation tile
constexpr int PMAX = 32;
constexpr float COS_EPS = 1e-8f;

struct Box { int x0, y0, x1, y1; };

__device__ __forceinline__ Box load_box(const int32_t* bp, int g, int Hp, int Wp) {
  Box b;
  b.x0 = max(bp[g * 4 + 0], 0); b.y0 = max(bp[g * 4 + 1], 0);
  b.x1 = min(bp[g * 4 + 2], Wp - 1); b.y1 = min(bp[g * 4 + 3], Hp - 1);
  return b;
}
__device__ __forceinline__ int box_w(const Box& b) { return max(b.x1 - b.x0 + 1, 0); }
__device__ __forceinline__ int box_count(const Box& b) { return box_w(b) * max(b.y1 - b.y0 + 1, 0); }
__device__ __forceinline__ int box_patch(const Box& b, int t, int Wp) {
  const int bw = box_w(b);
  const int ty = t / bw;
  return (b.y0 + ty) * Wp + b.x0 + (t - ty * bw);
}
__device__ __forceinline__ bool in_box(const Box& b, int n, int Wp) {
  const int y = n / Wp, x = n - y * Wp;
  return x >= b.x0 && x <= b.x1 && y >= b.y0 && y <= b.y1;
}

// 1 / max(||row||, eps) for `rows` rows of length C; one wave per row
__global__ __launch_bounds__(CS_NT) void row_invnorm_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                            int rows, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = x + (size_t)row * C;
  float s = 0.0f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(p + c);
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = 1.0f / fmaxf(sqrtf(s), COS_EPS);
}

// ---- sim pass ------------------------------------------------------------------------------------
// grid (tiles, G).  FULL=false: tile indexes the object's in-box patches; FULL=true: the whole grid.
template <bool FULL>
__global__ __launch_bounds__(CS_NT) void sim_kernel(const float* __restrict__ feat, const float* __restrict__ invn,
                                                    const float* __restrict__ prot, const float* __restrict__ invnp,
                                                    const int32_t* __restrict__ box_patch_,
                                                    const int32_t* __restrict__ obj_img,
                                                    const int32_t* __restrict__ assign_prev,   // [G,Np] or null
                                                    float* __restrict__ sim, float* __restrict__ part_stats,
                                                    int C, int Hp, int Wp, int P, int nt1) {
  __shared__ float red[4][32][33];
  const int Np = Hp * Wp;
  const int g = blockIdx.y, tile = blockIdx.x;
  const Box ob = load_box(box_patch_, g, Hp, Wp);
  const int nb = FULL ? Np : box_count(ob);
  if (tile * CS_TILE1 >= nb) return;
  const int b = obj_img[g];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;

  auto patch_of = [&](int local) {
    const int t = min(tile * CS_TILE1 + local, nb - 1);
    return FULL ? t : box_patch(ob, t, Wp);
  };
  const int n_mine = patch_of(li);
  const float* frow = feat + ((size_t)b * Np + n_mine) * C;
  const float* prow = prot + ((size_t)g * P + min(li, P - 1)) * C;
  const bool pvalid = li < P;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int nsteps = (C + 15) / 16;
  // SU k16 steps of this wave per trip, all 2*SU operand loads issued before the first MFMA (a step past the end
  // or a row past P reads a valid address and is multiplied by 0)
  constexpr int SU = 6;
  for (int s = wave; s < nsteps; s += 4 * SU) {
    Frag<float> fa[SU], fb[SU];
    float keep[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int su = s + 4 * u;
      const bool ok = su < nsteps && su * 16 + half * 8 + 8 <= C;
      const int k0 = ok ? su * 16 + half * 8 : 0;
      fb[u].load16B(frow + k0);
      fa[u].load16B(prow + k0);
      keep[u] = (ok && pvalid) ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
#pragma unroll
      for (int t = 0; t < 8; ++t) fa[u].v[t] *= keep[u];
      acc = mma32(fa[u], fb[u], acc);          // D[p][n]
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][acc_row(r, half)][li] = acc[r];
  __syncthreads();

  const int nn = tid & 31, pq = tid >> 5;
  const int t_loc = tile * CS_TILE1 + nn;
  const bool nvalid = t_loc < nb;
  const int n = patch_of(nn);
  const float fin = invn[(size_t)b * Np + n];
  const bool dens_ok = assign_prev != nullptr && nvalid && (!FULL || in_box(ob, n, Wp));
  const int a_prev = dens_ok ? assign_prev[(size_t)g * Np + n] : -1;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int p = pq + 8 * qd;
    float v = ((red[0][p][nn] + red[1][p][nn]) + red[2][p][nn]) + red[3][p][nn];
    v = v * (p < P ? invnp[g * PMAX + p] : 0.0f) * fin;
    if (p < P && nvalid) sim[((size_t)g * P + p) * Np + n] = v;
    float mx = (nvalid && p < P) ? v : -INFINITY;
    float ds = (a_prev == p) ? v : 0.0f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o));
      ds += __shfl_xor(ds, o);
    }
    if (nn == 0) {
      float* ps = part_stats + (((size_t)g * nt1 + tile) * PMAX + p) * 2;
      ps[0] = mx;
      ps[1] = ds;
    }
  }
}

// ---- stats pass: grid (P, G) ------------------------------------------------------------------------
// stats[g][p] = {tt = temp*tau, mlog = max logit, Z, tau}
__global__ __launch_bounds__(CS_NT) void stats_kernel(const float* __restrict__ sim,
                                                      const float* __restrict__ part_stats,
                                                      const int32_t* __restrict__ cnt,
                                                      const int32_t* __restrict__ box_patch_,
                                                      float* __restrict__ stats, float* __restrict__ tau_out,
                                                      float tau0, float temp, int it, int Hp, int Wp, int P, int G,
                                                      int nt1, int density_only) {
  __shared__ float sh_a[CS_NT], sh_b[CS_NT];
  const int Np = Hp * Wp;
  const int p = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const Box ob = load_box(box_patch_, g, Hp, Wp);
  const int nb = box_count(ob);
  const int ntiles = density_only ? (Np + CS_TILE1 - 1) / CS_TILE1 : (nb + CS_TILE1 - 1) / CS_TILE1;

  float mx = -INFINITY, ds = 0.0f;
  for (int t = tid; t < ntiles; t += CS_NT) {
    const float* ps = part_stats + (((size_t)g * nt1 + t) * PMAX + p) * 2;
    mx = fmaxf(mx, ps[0]);
    ds += ps[1];
  }
  sh_a[tid] = mx; sh_b[tid] = ds;
  __syncthreads();
  for (int o = CS_NT / 2; o > 0; o >>= 1) {
    if (tid < o) { sh_a[tid] = fmaxf(sh_a[tid], sh_a[tid + o]); sh_b[tid] += sh_b[tid + o]; }
    __syncthreads();
  }
  float maxsim = sh_a[0];
  const float dens = sh_b[0];
  __syncthreads();
  if (nb < Np) maxsim = fmaxf(maxsim, 0.0f);          // out-of-box patches have cosine exactly 0

  float tau = tau0;
  if (it > 0) {                                       // update_density_batch (:882-908)
    const float c = (float)cnt[g * PMAX + p];
    const float mean = c >= 1.0f ? dens / c : 0.0f;
    tau = fmaxf(1.0f - mean, 1e-10f);
    if (tau_out != nullptr && tid == 0) tau_out[((size_t)(it - 1) * G + g) * P + p] = tau;
  }
  if (density_only) return;
  const float tt = temp * tau;
  const float mlog = maxsim / tt;
  float z = 0.0f;
  const float* srow = sim + ((size_t)g * P + p) * Np;
  for (int t = tid; t < nb; t += CS_NT) z += expf(srow[box_patch(ob, t, Wp)] / tt - mlog);
  sh_a[tid] = z;
  __syncthreads();
  for (int o = CS_NT / 2; o > 0; o >>= 1) {
    if (tid < o) sh_a[tid] += sh_a[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    z = sh_a[0] + (float)(Np - nb) * expf(0.0f / tt - mlog);
    float* st = stats + ((size_t)g * PMAX + p) * 4;
    st[0] = tt; st[1] = mlog; st[2] = z; st[3] = tau;
  }
}

// argmax over prototypes of the softmax weight of an out-of-box (zero-feature) patch
__device__ __forceinline__ int outside_cluster(const float* st_g, int P) {
  int best = 0;
  float bw = -INFINITY;
  for (int p = 0; p < P; ++p) {
    const float w = expf(0.0f / st_g[p * 4 + 0] - st_g[p * 4 + 1]) / st_g[p * 4 + 2];
    if (w > bw) { bw = w; best = p; }
  }
  return best;
}

// ---- assign + aggregate: grid (nt3, G) ---------------------------------------------------------------
template <int CPT>   // channels per thread = ceil(C / 256)
__global__ __launch_bounds__(CS_NT) void assign_kernel(const float* __restrict__ feat, const float* __restrict__ sim,
                                                       const float* __restrict__ stats,
                                                       const int32_t* __restrict__ box_patch_,
                                                       const int32_t* __restrict__ obj_img,
                                                       int32_t* __restrict__ assign, int32_t* __restrict__ assign_out,
                                                       float* __restrict__ part_prot, int32_t* __restrict__ part_cnt,
                                                       int C, int Hp, int Wp, int P, int nt3) {
  __shared__ float st_s[PMAX * 4];
  __shared__ int a_s[CS_TILE3];
  __shared__ float w_s[CS_TILE3];
  __shared__ int n_s[CS_TILE3];
  __shared__ int cnt_s[PMAX];
  const int Np = Hp * Wp;
  const int g = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const Box ob = load_box(box_patch_, g, Hp, Wp);
  const int nb = box_count(ob);
  const int b = obj_img[g];
  if (tid < P * 4) st_s[tid] = stats[(size_t)g * PMAX * 4 + tid];
  if (tid < PMAX) cnt_s[tid] = 0;
  __syncthreads();

  // user-visible assignment of out-of-box patches (never read back by the iteration itself)
  if (assign_out != nullptr) {
    const int n = tile * CS_TILE3 + tid;
    if (tid < CS_TILE3 && n < Np && !in_box(ob, n, Wp)) assign_out[(size_t)g * Np + n] = outside_cluster(st_s, P);
  }
  const int t0 = tile * CS_TILE3;
  if (t0 >= nb) return;
  const int count = min(CS_TILE3, nb - t0);

  if (tid < count) {
    const int n = box_patch(ob, t0 + tid, Wp);
    int best = 0;
    float bw = -INFINITY;
    for (int p = 0; p < P; ++p) {
      const float w = expf(sim[((size_t)g * P + p) * Np + n] / st_s[p * 4 + 0] - st_s[p * 4 + 1]) / st_s[p * 4 + 2];
      if (w > bw) { bw = w; best = p; }           // strict: ties keep the lowest prototype index
    }
    a_s[tid] = best; w_s[tid] = bw; n_s[tid] = n;
    assign[(size_t)g * Np + n] = best;
    if (assign_out != nullptr) assign_out[(size_t)g * Np + n] = best;
    atomicAdd(&cnt_s[best], 1);
  }
  __syncthreads();

  float acc[CPT][PMAX];
#pragma unroll
  for (int i = 0; i < CPT; ++i)
#pragma unroll
    for (int p = 0; p < PMAX; ++p) acc[i][p] = 0.0f;
  const float* fb = feat + (size_t)b * Np * C;
  constexpr int UNR = 8;                 // feature rows of UNR patches are in flight before the first one is used
  for (int j0 = 0; j0 < count; j0 += UNR) {
    float f[UNR][CPT];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float* frow = fb + (size_t)n_s[min(j0 + u, count - 1)] * C;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int c = tid + i * CS_NT;
        f[u][i] = c < C ? frow[c] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (j0 + u < count) {                // workgroup-uniform
        const int a = __builtin_amdgcn_readfirstlane(a_s[j0 + u]);
        const float wa = w_s[j0 + u];
        // wave-uniform selection of the accumulator: compiles to a scalar branch tree, registers stay static
#define AS_CASE(PP)                                                            \
  case PP:                                                                     \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) acc[i][PP] = fmaf(wa, f[u][i], acc[i][PP]); \
    break;
        switch (a) {
          AS_CASE(0) AS_CASE(1) AS_CASE(2) AS_CASE(3) AS_CASE(4) AS_CASE(5) AS_CASE(6) AS_CASE(7)
          AS_CASE(8) AS_CASE(9) AS_CASE(10) AS_CASE(11) AS_CASE(12) AS_CASE(13) AS_CASE(14) AS_CASE(15)
          AS_CASE(16) AS_CASE(17) AS_CASE(18) AS_CASE(19) AS_CASE(20) AS_CASE(21) AS_CASE(22) AS_CASE(23)
          AS_CASE(24) AS_CASE(25) AS_CASE(26) AS_CASE(27) AS_CASE(28) AS_CASE(29) AS_CASE(30) AS_CASE(31)
          default: break;
        }
#undef AS_CASE
      }
    }
  }
  float* pp = part_prot + ((size_t)g * nt3 + tile) * P * C;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) {
    if (p < P) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int c = tid + i * CS_NT;
        if (c < C) pp[(size_t)p * C + c] = acc[i][p];
      }
    }
  }
  if (tid < PMAX) part_cnt[((size_t)g * nt3 + tile) * PMAX + tid] = cnt_s[tid];
}

// ---- finalize: grid (P, G): sum partial prototypes in tile order, norm, member count -----------------
__global__ __launch_bounds__(CS_NT) void finalize_kernel(const float* __restrict__ part_prot,
                                                         const int32_t* __restrict__ part_cnt,
                                                         const float* __restrict__ stats,
                                                         const int32_t* __restrict__ box_patch_,
                                                         float* __restrict__ prot, float* __restrict__ invnp,
                                                         int32_t* __restrict__ cnt, int C, int Hp, int Wp, int P,
                                                         int nt3) {
  __shared__ float sh[CS_NT];
  const int Np = Hp * Wp;
  const int p = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const Box ob = load_box(box_patch_, g, Hp, Wp);
  const int nb = box_count(ob);
  const int ntiles = (nb + CS_TILE3 - 1) / CS_TILE3;
  float sq = 0.0f;
  for (int c = tid; c < C; c += CS_NT) {
    // four interleaved partial sums (independent loads in flight), combined in a fixed order: deterministic
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
    const float* base = part_prot + (((size_t)g * nt3) * P + p) * C + c;
    const size_t step = (size_t)P * C;
    int t = 0;
    for (; t + 4 <= ntiles; t += 4) {
      v0 += base[(size_t)t * step];
      v1 += base[(size_t)(t + 1) * step];
      v2 += base[(size_t)(t + 2) * step];
      v3 += base[(size_t)(t + 3) * step];
    }
    for (; t < ntiles; ++t) v0 += base[(size_t)t * step];
    const float v = (v0 + v1) + (v2 + v3);
    prot[((size_t)g * P + p) * C + c] = v;
    sq = fmaf(v, v, sq);
  }
  sh[tid] = sq;
  __syncthreads();
  for (int o = CS_NT / 2; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    invnp[g * PMAX + p] = 1.0f / fmaxf(sqrtf(sh[0]), COS_EPS);
    int c = 0;
    for (int t = 0; t < ntiles; ++t) c += part_cnt[((size_t)g * nt3 + t) * PMAX + p];
    if (nb < Np && outside_cluster(stats + (size_t)g * PMAX * 4, P) == p) c += Np - nb;
    cnt[g * PMAX + p] = c;
  }
}

__global__ void prot_invnorm_kernel(const float* __restrict__ prot, float* __restrict__ invnp, int G, int P, int C) {
  const int gp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (gp >= G * P) return;
  const float* r = prot + (size_t)gp * C;
  float s = 0.0f;
  for (int c = lane; c < C; c += 64) s = fmaf(r[c], r[c], s);
  s = wave_sum(s);
  if (lane == 0) invnp[(gp / P) * PMAX + gp % P] = 1.0f / fmaxf(sqrtf(s), COS_EPS);
}

struct WsLayout {
  size_t invn, invnp, cnt, stats, part_stats, assign, part_prot, part_cnt, total;
  int nt1, nt3;
};
WsLayout ws_layout(int B, int C, int Np, int G, int P) {
  WsLayout w;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  w.nt1 = as_ceil_div(Np, CS_TILE1);
  w.nt3 = as_ceil_div(Np, CS_TILE3);
  size_t o = 0;
  w.invn = o; o = al(o + (size_t)B * Np * 4);
  w.invnp = o; o = al(o + (size_t)G * PMAX * 4);
  w.cnt = o; o = al(o + (size_t)G * PMAX * 4);
  w.stats = o; o = al(o + (size_t)G * PMAX * 16);
  w.part_stats = o; o = al(o + (size_t)G * w.nt1 * PMAX * 8);
  w.assign = o; o = al(o + (size_t)G * Np * 4);
  w.part_prot = o; o = al(o + (size_t)G * w.nt3 * P * C * 4);
  w.part_cnt = o; o = al(o + (size_t)G * w.nt3 * PMAX * 4);
  w.total = o;
  return w;
}

}  // namespace

extern "C" size_t as_cosine_shift_workspace_bytes(int B, int C, int Hp, int Wp, int G, int P) {
  if (B <= 0 || C <= 0 || Hp <= 0 || Wp <= 0 || G <= 0 || P <= 0) return 0;
  return ws_layout(B, C, Hp * Wp, G, P).total;
}

extern "C" int as_cosine_shift(const float* feat, const int32_t* box_patch, const int32_t* obj_img, float* prot,
                               float tau0, float temp, int n_shift, float* sim_out, int32_t* assign_out,
                               float* tau_out, void* ws, size_t ws_bytes, int B, int C, int Hp, int Wp, int G, int P,
                               as_stream_t stream) {
  AS_REQUIRE(feat && box_patch && obj_img && prot && sim_out && ws, AS_E_BADARG, "as_cosine_shift: null pointer");
  AS_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && G > 0 && n_shift >= 0, AS_E_BADARG, "as_cosine_shift: bad sizes");
  AS_REQUIRE(P > 0 && P <= PMAX, AS_E_UNSUPPORTED, "as_cosine_shift: P=%d prototypes per object (max %d)", P, PMAX);
  AS_REQUIRE(C % 8 == 0 && C <= 4 * CS_NT, AS_E_UNSUPPORTED, "as_cosine_shift: C=%d must be a multiple of 8 and <= 1024", C);
  const int Np = Hp * Wp;
  const WsLayout L = ws_layout(B, C, Np, G, P);
  AS_REQUIRE(ws_bytes >= L.total, AS_E_WORKSPACE, "as_cosine_shift: workspace %zu < %zu bytes", ws_bytes, L.total);
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  float* invn = (float*)(w + L.invn);
  float* invnp = (float*)(w + L.invnp);
  int32_t* cnt = (int32_t*)(w + L.cnt);
  float* stats = (float*)(w + L.stats);
  float* part_stats = (float*)(w + L.part_stats);
  int32_t* assign = (int32_t*)(w + L.assign);
  float* part_prot = (float*)(w + L.part_prot);
  int32_t* part_cnt = (int32_t*)(w + L.part_cnt);

  hipLaunchKernelGGL(row_invnorm_kernel, dim3(as_ceil_div(B * Np, 4)), dim3(CS_NT), 0, s, feat, invn, B * Np, C);
  hipLaunchKernelGGL(prot_invnorm_kernel, dim3(as_ceil_div(G * P, 4)), dim3(CS_NT), 0, s, prot, invnp, G, P, C);
  const int cpt = as_ceil_div(C, CS_NT);
  for (int it = 0; it < n_shift; ++it) {
    hipLaunchKernelGGL((sim_kernel<false>), dim3(L.nt1, G), dim3(CS_NT), 0, s, feat, invn, prot, invnp, box_patch,
                       obj_img, it > 0 ? assign : nullptr, sim_out, part_stats, C, Hp, Wp, P, L.nt1);
    hipLaunchKernelGGL(stats_kernel, dim3(P, G), dim3(CS_NT), 0, s, sim_out, part_stats, cnt, box_patch, stats,
                       tau_out, tau0, temp, it, Hp, Wp, P, G, L.nt1, 0);
    int32_t* aout = assign_out ? assign_out + (size_t)it * G * Np : nullptr;
#define AS_ASSIGN(CPT)                                                                                         \
  hipLaunchKernelGGL((assign_kernel<CPT>), dim3(L.nt3, G), dim3(CS_NT), 0, s, feat, sim_out, stats, box_patch, \
                     obj_img, assign, aout, part_prot, part_cnt, C, Hp, Wp, P, L.nt3)
    switch (cpt) {
      case 1: AS_ASSIGN(1); break;
      case 2: AS_ASSIGN(2); break;
      case 3: AS_ASSIGN(3); break;
      default: AS_ASSIGN(4); break;
    }
#undef AS_ASSIGN
    hipLaunchKernelGGL(finalize_kernel, dim3(P, G), dim3(CS_NT), 0, s, part_prot, part_cnt, stats, box_patch, prot,
                       invnp, cnt, C, Hp, Wp, P, L.nt3);
  }
  // final similarity on the UNMASKED map (+ the density of the last assignment for tau_out)
  hipLaunchKernelGGL((sim_kernel<true>), dim3(L.nt1, G), dim3(CS_NT), 0, s, feat, invn, prot, invnp, box_patch,
                     obj_img, n_shift > 0 ? assign : nullptr, sim_out, part_stats, C, Hp, Wp, P, L.nt1);
  if (n_shift > 0 && tau_out != nullptr)
    hipLaunchKernelGGL(stats_kernel, dim3(P, G), dim3(CS_NT), 0, s, sim_out, part_stats, cnt, box_patch, stats,
                       tau_out, tau0, temp, n_shift, Hp, Wp, P, G, L.nt1, 1);
  AS_CHECK_LAUNCH("cosine_shift");
  return AS_OK;
}

// =====================================================================================================
// Cosine-affinity refinement on the patch grid (stdroi:668-707 get_refined_similarity), one image.
// Reuses the exact-fp32 MFMA similarity pass above with the seeds as the "prototypes" of one object
// whose box is the whole grid.
// =====================================================================================================
namespace {

__global__ void refine_setup_kernel(int32_t* ints, int Hp, int Wp) {
  if (threadIdx.x == 0) { ints[0] = 0; ints[1] = 0; ints[2] = Wp - 1; ints[3] = Hp - 1; ints[4] = 0; }
}

// level output: optional box masking of the first G maps + keep-the-winner selection (:676-683, :697-703)
__global__ __launch_bounds__(CS_NT) void refine_select_kernel(float* __restrict__ work, const int32_t* __restrict__ boxes,
                                                              float* __restrict__ out, int G, int Gp, int Hp, int Wp,
                                                              int is_select, int write_back) {
  const int Np = Hp * Wp;
  const int n = blockIdx.x * CS_NT + threadIdx.x;
  if (n >= Np) return;
  if (!is_select) {
    for (int g = 0; g < Gp; ++g) out[(size_t)g * Np + n] = work[(size_t)g * Np + n];
    return;
  }
  int best = 0;
  float bv = -INFINITY;
  for (int g = 0; g < Gp; ++g) {
    float v = work[(size_t)g * Np + n];
    if (g < G) {
      const Box bx = load_box(boxes, g, Hp, Wp);
      v = v * (in_box(bx, n, Wp) ? 1.0f : 0.0f);
      if (write_back) work[(size_t)g * Np + n] = v;
    }
    if (v > bv) { bv = v; best = g; }
  }
  for (int g = 0; g < Gp; ++g) {
    float v = work[(size_t)g * Np + n];
    if (g < G && !write_back) {
      const Box bx = load_box(boxes, g, Hp, Wp);
      v = v * (in_box(bx, n, Wp) ? 1.0f : 0.0f);
    }
    out[(size_t)g * Np + n] = (g == best) ? v : 0.0f;
  }
}

// threshold at tau * rowmax, similarity-weighted mean feature (:687-691), in three deterministic steps:
//   rowmax   grid (Gp)          peak[g] = max_n work[g][n]
//   partial  grid (tiles, Gp)   partial[g][tile][:] = sum over the tile's surviving patches of w * feat[n][:]
//   finish   grid (Gp)          seeds_out[g] = sum_tiles partial / clamp(sum w, 1e-8)
constexpr int RF_TILE = 128;

__global__ __launch_bounds__(CS_NT) void refine_rowmax_kernel(const float* __restrict__ work, float* __restrict__ peak,
                                                              int Np) {
  __shared__ float sh[CS_NT];
  const int g = blockIdx.x, tid = threadIdx.x;
  float mx = -INFINITY;
  for (int n = tid; n < Np; n += CS_NT) mx = fmaxf(mx, work[(size_t)g * Np + n]);
  sh[tid] = mx;
  __syncthreads();
  for (int o = CS_NT / 2; o > 0; o >>= 1) {
    if (tid < o) sh[tid] = fmaxf(sh[tid], sh[tid + o]);
    __syncthreads();
  }
  if (tid == 0) peak[g] = sh[0];
}

template <int CPT>
__global__ __launch_bounds__(CS_NT) void refine_partial_kernel(const float* __restrict__ feat,
                                                               const float* __restrict__ work,
                                                               const float* __restrict__ peak, float tau,
                                                               float* __restrict__ partial, float* __restrict__ partial_w,
                                                               int C, int Np, int ntile) {
  __shared__ float w_s[RF_TILE];
  const int tile = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int n0 = tile * RF_TILE;
  const int count = min(RF_TILE, Np - n0);
  const float thr = peak[g] * tau;
  if (tid < RF_TILE) {
    float w = 0.0f;
    if (tid < count) { w = work[(size_t)g * Np + n0 + tid]; if (w < thr) w = 0.0f; }   // cos_map1[cos_map1 < thr] *= 0
    w_s[tid] = w;
  }
  __syncthreads();
  // surviving patches, in patch order (thread 0 compacts: <= RF_TILE entries), then the weighted sum with 8 feature
  // rows in flight per trip -- the same order of fma's as a plain loop, without one exposed load latency per patch
  __shared__ int j_s[RF_TILE];
  __shared__ int nlive_s;
  if (tid == 0) {
    int nl = 0;
    for (int j = 0; j < count; ++j)
      if (w_s[j] != 0.0f) j_s[nl++] = j;
    nlive_s = nl;
  }
  __syncthreads();
  const int nlive = nlive_s;
  float acc[CPT], wsum = 0.0f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) acc[i] = 0.0f;
  constexpr int UNR = 8;
  for (int q0 = 0; q0 < nlive; q0 += UNR) {
    float f[UNR][CPT];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float* frow = feat + (size_t)(n0 + j_s[min(q0 + u, nlive - 1)]) * C;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int c = tid + i * CS_NT;
        f[u][i] = c < C ? frow[c] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (q0 + u < nlive) {                        // workgroup-uniform
        const float w = w_s[j_s[q0 + u]];
        wsum += w;
#pragma unroll
        for (int i = 0; i < CPT; ++i) acc[i] = fmaf(w, f[u][i], acc[i]);
      }
    }
  }
  float* pp = partial + ((size_t)g * ntile + tile) * C;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + i * CS_NT;
    if (c < C) pp[c] = acc[i];
  }
  if (tid == 0) partial_w[g * ntile + tile] = wsum;
}

__global__ __launch_bounds__(CS_NT) void refine_finish_kernel(const float* __restrict__ partial,
                                                              const float* __restrict__ partial_w,
                                                              float* __restrict__ seeds_out, int C, int ntile) {
  const int g = blockIdx.x, tid = threadIdx.x;
  float wsum = 0.0f;
  for (int t = 0; t < ntile; ++t) wsum += partial_w[g * ntile + t];
  const float den = fmaxf(wsum, 1e-8f);
  for (int c = tid; c < C; c += CS_NT) {
    float v = 0.0f;
    for (int t = 0; t < ntile; ++t) v += partial[((size_t)g * ntile + t) * C + c];
    seeds_out[(size_t)g * C + c] = v / den;
  }
}

struct RefineWs { size_t invn, invnp, ints, part_stats, work, peak, partial, partial_w, total; int nt1, ntile; };
RefineWs refine_ws(int C, int Np, int Gp) {
  RefineWs w;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  w.nt1 = as_ceil_div(Np, CS_TILE1);
  size_t o = 0;
  w.invn = o; o = al(o + (size_t)Np * 4);
  w.invnp = o; o = al(o + (size_t)PMAX * 4);
  w.ints = o; o = al(o + 64);
  w.part_stats = o; o = al(o + (size_t)w.nt1 * PMAX * 8);
  w.work = o; o = al(o + (size_t)Gp * Np * 4);
  w.ntile = as_ceil_div(Np, RF_TILE);
  w.peak = o; o = al(o + (size_t)PMAX * 4);
  w.partial = o; o = al(o + (size_t)Gp * w.ntile * C * 4);
  w.partial_w = o; o = al(o + (size_t)Gp * w.ntile * 4);
  w.total = o;
  return w;
}

}  // namespace

extern "C" size_t as_refine_similarity_workspace_bytes(int C, int Np, int Gp) {
  if (C <= 0 || Np <= 0 || Gp <= 0) return 0;
  return refine_ws(C, Np, Gp).total;
}

extern "C" int as_refine_similarity(const float* feat, const float* seeds, const int32_t* boxes, int G, int Gp,
                                    int refine_times, float tau, int is_select, float* maps, float* seeds_out, void* ws,
                                    size_t ws_bytes, int C, int Hp, int Wp, as_stream_t stream) {
  AS_REQUIRE(feat && seeds && maps && seeds_out && ws && (boxes || !is_select), AS_E_BADARG, "as_refine_similarity: null pointer");
  AS_REQUIRE(G >= 0 && Gp > 0 && G <= Gp && Gp <= PMAX, AS_E_UNSUPPORTED, "as_refine_similarity: Gp=%d seeds (max %d)", Gp, PMAX);
  AS_REQUIRE(C % 8 == 0 && C <= 4 * CS_NT && Hp > 0 && Wp > 0 && refine_times >= 0, AS_E_UNSUPPORTED,
             "as_refine_similarity: C=%d must be a multiple of 8 and <= 1024", C);
  const int Np = Hp * Wp;
  const RefineWs L = refine_ws(C, Np, Gp);
  AS_REQUIRE(ws_bytes >= L.total, AS_E_WORKSPACE, "as_refine_similarity: workspace %zu < %zu bytes", ws_bytes, L.total);
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  float* invn = (float*)(w + L.invn);
  float* invnp = (float*)(w + L.invnp);
  int32_t* ints = (int32_t*)(w + L.ints);
  float* part_stats = (float*)(w + L.part_stats);
  float* work = (float*)(w + L.work);
  float* peak = (float*)(w + L.peak);
  float* partial = (float*)(w + L.partial);
  float* partial_w = (float*)(w + L.partial_w);
  const int cpt = as_ceil_div(C, CS_NT);

  hipLaunchKernelGGL(refine_setup_kernel, dim3(1), dim3(64), 0, s, ints, Hp, Wp);
  hipLaunchKernelGGL(row_invnorm_kernel, dim3(as_ceil_div(Np, 4)), dim3(CS_NT), 0, s, feat, invn, Np, C);
  const float* cur = seeds;
  for (int lvl = 0; lvl <= refine_times; ++lvl) {
    if (lvl > 0) {
#define AS_AGG(CPT)                                                                                          \
  hipLaunchKernelGGL((refine_partial_kernel<CPT>), dim3(L.ntile, Gp), dim3(CS_NT), 0, s, feat, work, peak, tau, \
                     partial, partial_w, C, Np, L.ntile)
      hipLaunchKernelGGL(refine_rowmax_kernel, dim3(Gp), dim3(CS_NT), 0, s, work, peak, Np);
      switch (cpt) {
        case 1: AS_AGG(1); break;
        case 2: AS_AGG(2); break;
        case 3: AS_AGG(3); break;
        default: AS_AGG(4); break;
      }
      hipLaunchKernelGGL(refine_finish_kernel, dim3(Gp), dim3(CS_NT), 0, s, partial, partial_w, seeds_out, C, L.ntile);
#undef AS_AGG
      cur = seeds_out;
    }
    hipLaunchKernelGGL(prot_invnorm_kernel, dim3(as_ceil_div(Gp, 4)), dim3(CS_NT), 0, s, cur, invnp, 1, Gp, C);
    hipLaunchKernelGGL((sim_kernel<true>), dim3(L.nt1, 1), dim3(CS_NT), 0, s, feat, invn, cur, invnp, ints, ints + 4,
                       (const int32_t*)nullptr, work, part_stats, C, Hp, Wp, Gp, L.nt1);
    hipLaunchKernelGGL(refine_select_kernel, dim3(as_ceil_div(Np, CS_NT)), dim3(CS_NT), 0, s, work, boxes,
                       maps + (size_t)lvl * Gp * Np, G, Gp, Hp, Wp, is_select, lvl > 0 ? 1 : 0);
  }
  if (refine_times == 0)
    (void)hipMemcpyAsync(seeds_out, seeds, (size_t)Gp * C * 4, hipMemcpyDeviceToDevice, s);
  AS_CHECK_LAUNCH("refine_similarity");
  return AS_OK;
}
