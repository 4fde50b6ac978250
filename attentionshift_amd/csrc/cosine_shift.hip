// Mean-shift token clustering ("attention shift" iteration) for gfx950, fp32 throughout.
//
// Restates cosine_shift_batch + update_density_batch
// (reference mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py:830-854, 882-908) with the
// box masking of mean_shift_grid_prototype (:1819-1824) folded in: features outside an object's box are
// zero in the reference, so their cosine is exactly 0, they add exp(-max) to every softmax denominator,
// nothing to the aggregation, and all fall into ONE cluster (the argmax of exp(-max_p)/Z_p).  The loop
// therefore only touches in-box patches; the final similarity map is over the whole (unmasked) grid.
//
// Per iteration, three launches (deterministic, no float atomics, every sum in a fixed order):
//   sim       : cos(prot, feat) tiles, 32 in-box patches x <=32 prototypes, exact-fp32 MFMA
//               (v_mfma_f32_32x32x2), the 8 waves of a workgroup split the channel range; also per-tile max
//               and the density sums of the PREVIOUS assignment (update_density_batch needs
//               cos(new prot, feat) = this pass).  Similarities are kept compact ([g][p][in-box index]).
//   assign    : every tile workgroup rebuilds its object's statistics (tau, logit max, Z) from the tile
//               partials and the compact rows, then assigns its 32 patches:
//               w = exp(sim/(temp*tau) - max)/Z, argmax over prototypes (ties -> lowest)
//   aggregate : prot_new[a] += w * feat[n], owned per 32-channel block of ALL prototypes of an object, so there
//               are no cross-workgroup partial prototypes; norms and member counts come out of the same pass
// These kernels are latency-bound (each dependent global-memory hop of a small grid costs ~2 us): operands are
// requested before the box is known, and every batch of loads is fenced (sched_barrier) against the scheduler
// sinking the loads to their uses.
#include "common.h"
#include "bilinear.h"

namespace {

constexpr int CS_NT = 256;
constexpr int CS_TILE1 = 32;     // patches per sim tile
constexpr int PMAX = 32;
constexpr float COS_EPS = 1e-8f;

struct Box { int x0, y0, x1, y1; };

// -DAS_SHIFT_STAMPS (tools/experiments/shift_timeline.py): thread 0 of every workgroup writes s_memrealtime (100 MHz, one
// counter for the whole chip) at the phase boundaries of the three iteration kernels; `drain` first waits for the
// workgroup's outstanding loads so that the stamp is "operands landed", not "operands requested".
#ifdef AS_SHIFT_STAMPS
constexpr int ST_SLOTS = 8, ST_BLOCKS = 1024, ST_LAUNCH = 16;
__device__ unsigned long long g_stamps[ST_LAUNCH][ST_BLOCKS][ST_SLOTS];
#define AS_STAMP_ARG , int stamp_id
#define AS_STAMP_VAL(x) , x
#define AS_STAMP(slot, drain)                                                                                      \
  do {                                                                                                             \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
    if (threadIdx.x == 0) g_stamps[stamp_id][blockIdx.y * gridDim.x + blockIdx.x][slot] = wall_clock64();           \
  } while (0)
#else
#define AS_STAMP_ARG
#define AS_STAMP_VAL(x)
#define AS_STAMP(slot, drain)
#endif

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

// ---- stats pass: grid (P, G) ------------------------------------------------------------------------
// stats[g][p] = {tt = temp*tau, mlog = max logit, Z, tau}
__global__ __launch_bounds__(CS_NT) void stats_kernel(const float* __restrict__ sim,
                                                      const float* __restrict__ part_stats,
                                                      const int32_t* __restrict__ cnt,
                                                      const int32_t* __restrict__ box_patch_,
                                                      float* __restrict__ stats, float* __restrict__ tau_out,
                                                      float tau0, float temp, float tt0, int it, int Hp, int Wp, int P,
                                                      int G, int nt1, int density_only) {
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
  const float tt = it == 0 ? tt0 : temp * tau;
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

// ---- shift iteration, pass 1: similarity tiles (8 waves split the channel range) ---------------------
// grid (tiles, G): tile indexes the object's in-box patches, sim is written COMPACT ([g][p][in-box index]).
constexpr int S1_NT = 512;
constexpr int SH_CH = 32;        // channels per aggregation workgroup

// The patch norms and (first iteration) the prototype norms are accumulated from the MFMA operand fragments, so there is
// no norm pass over the feature map; later iterations sum the `npart` squared-norm partials of the aggregation pass with
// one lane per partial and a fixed shuffle tree.
__global__ __launch_bounds__(S1_NT) void shift_sim_kernel(const float* __restrict__ feat, const float* __restrict__ prot,
                                                          const float* __restrict__ pn2, int npart,
                                                          const int32_t* __restrict__ box_patch_,
                                                          const int32_t* __restrict__ obj_img,
                                                          const int2* __restrict__ aw_prev,   // [G][Np] compact or null
                                                          float* __restrict__ sim, float* __restrict__ part_stats,
                                                          int C, int Hp, int Wp, int P, int nt1, size_t fbs AS_STAMP_ARG) {
  __shared__ float red[8][32][33];
  __shared__ float nrm[2][16][32];
  __shared__ float invnp_s[PMAX];
  const int Np = Hp * Wp;
  const int g = blockIdx.y, tile = blockIdx.x;
  AS_STAMP(0, false);
  const Box ob = load_box(box_patch_, g, Hp, Wp);
  const int nb = box_count(ob);
  if (tile * CS_TILE1 >= nb) return;
  AS_STAMP(1, false);                            // box known
  const int b = obj_img[g];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, half = lane >> 5;

  if (pn2 != nullptr) {                         // |prot_p|^2 = sum of npart (<= 64) partials: wave w owns p = w, w+8, ...
    for (int p = wave; p < PMAX; p += 8) {
      float s = (p < P && lane < npart) ? pn2[((size_t)g * npart + lane) * PMAX + p] : 0.0f;
      s = wave_sum(s);
      if (lane == 0) invnp_s[p] = p < P ? 1.0f / fmaxf(sqrtf(s), COS_EPS) : 0.0f;
    }
  }

  auto patch_of = [&](int local) { return box_patch(ob, min(tile * CS_TILE1 + local, nb - 1), Wp); };
  const int n_mine = patch_of(li);
  const float* frow = feat + (size_t)b * fbs + (size_t)n_mine * C;      // fbs: floats between two images' token blocks
  const float* prow = prot + ((size_t)g * P + min(li, P - 1)) * C;
  const bool pvalid = li < P;

  // epilogue operands requested before the contraction: this thread's (patch nn, prototype rows pq, pq+16)
  const int nn = tid & 31, pq = tid >> 5;
  const int t_loc = tile * CS_TILE1 + nn;
  const bool nvalid = t_loc < nb;
  int a_prev = -1;
  if (aw_prev != nullptr && nvalid) a_prev = aw_prev[(size_t)g * Np + t_loc].x;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float qa = 0.0f, qb = 0.0f;                    // squared-norm partials of this lane's operand fragments
  const int nsteps = (C + 15) / 16;
  constexpr int SU = 6;
  for (int s = wave; s < nsteps; s += 8 * SU) {
    Frag<float> fa[SU], fb[SU];
    float keep[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int su = s + 8 * u;
      const bool ok = su < nsteps && su * 16 + half * 8 + 8 <= C;
      const int k0 = ok ? su * 16 + half * 8 : 0;
      fb[u].load16B(frow + k0);
      fa[u].load16B(prow + k0);
      keep[u] = ok ? 1.0f : 0.0f;
    }
    // keep all 4*SU operand loads in flight: without the fence the scheduler sinks each load to its MFMA to save
    // registers and the loop degenerates into SU dependent memory round trips
    __builtin_amdgcn_sched_barrier(0);
    AS_STAMP(2, true);                           // operand fragments landed
#pragma unroll
    for (int u = 0; u < SU; ++u) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        fa[u].v[t] *= pvalid ? keep[u] : 0.0f;
        fb[u].v[t] *= keep[u];
        qa = fmaf(fa[u].v[t], fa[u].v[t], qa);
        qb = fmaf(fb[u].v[t], fb[u].v[t], qb);
      }
      acc = mma32(fa[u], fb[u], acc);          // D[p][n]
    }
  }
  AS_STAMP(3, false);                            // MFMA chain issued
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][acc_row(r, half)][li] = acc[r];
  nrm[0][wave * 2 + half][li] = qa;
  nrm[1][wave * 2 + half][li] = qb;
  __syncthreads();
  AS_STAMP(4, false);                            // partials exchanged

  float fn = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) fn += nrm[1][k][nn];
  const float fin = 1.0f / fmaxf(sqrtf(fn), COS_EPS);
#pragma unroll
  for (int qd = 0; qd < 2; ++qd) {
    const int p = pq + 16 * qd;
    float pin;
    if (pn2 != nullptr) {
      pin = invnp_s[p];
    } else {
      float sn = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) sn += nrm[0][k][p];
      pin = 1.0f / fmaxf(sqrtf(sn), COS_EPS);
    }
    float v = (((red[0][p][nn] + red[1][p][nn]) + (red[2][p][nn] + red[3][p][nn])) +
               ((red[4][p][nn] + red[5][p][nn]) + (red[6][p][nn] + red[7][p][nn])));
    v = v * pin * fin;
    if (p < P && nvalid) sim[((size_t)g * P + p) * Np + t_loc] = v;
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
  AS_STAMP(5, false);                            // results stored (issued)
}

// ---- shift iteration, pass 2: per-prototype softmax statistics + assignment of this tile ------------------
// grid (tiles, G).  Every tile workgroup of an object recomputes the object's statistics (tau, logit max, Z over
// the compact similarity rows: a few thousand exps) instead of waiting for a separate statistics launch; it then
// assigns its 32 patches: w = exp(sim/(temp*tau) - max)/Z, strict-> argmax over prototypes (ties -> lowest).
__global__ __launch_bounds__(CS_NT) void shift_assign_kernel(const float* __restrict__ sim_c,
                                                             const float* __restrict__ part_stats,
                                                             const int32_t* __restrict__ cnt,
                                                             const int32_t* __restrict__ box_patch_,
                                                             float* __restrict__ stats, float* __restrict__ tau_out,
                                                             int2* __restrict__ aw, int32_t* __restrict__ assign_out,
                                                             int32_t* __restrict__ oc_out,
                                                             float tau0, float temp, float tt0, int it, int Hp, int Wp,
                                                             int P, int G, int nt1 AS_STAMP_ARG) {
  __shared__ float sh_mx[8][32], sh_ds[8][32];
  __shared__ float st_s[PMAX * 4], zs_s[PMAX * 2];
  const int Np = Hp * Wp;
  const int g = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int t0 = tile * CS_TILE1;
  AS_STAMP(0, false);
  // Every dependent global-memory hop of these small kernels costs ~2 us, so ALL operands are requested before
  // anything is known about the box: addresses are clamped to the array, masks are applied when the box arrives.
  constexpr int PSU = 8;                       // part_stats tiles per thread per trip (8 slots x PSU = 64 tiles)
  constexpr int ZQ = PMAX / 4, ZU = 12;        // Z trip: ZQ prototypes x ZU patches per lane (768 patches)
  const int sp = tid & 31, slot = tid >> 5;
  float2 ps[PSU];
#pragma unroll
  for (int k = 0; k < PSU; ++k)
    ps[k] = *reinterpret_cast<const float2*>(part_stats + (((size_t)g * nt1 + min(slot + 8 * k, nt1 - 1)) * PMAX + sp) * 2);
  // (the bulk -- the similarity rows for Z -- is requested blind only by the first 32 tiles of an object, which
  // cover boxes up to 1024 patches; later tiles, mostly beyond the box, wait for the box first)
  const bool eager = tile < 32;
  float v[ZQ][ZU];
  if (eager) {
#pragma unroll
    for (int q = 0; q < ZQ; ++q) {
      const float* srow = sim_c + ((size_t)g * P + min(wave + 4 * q, P - 1)) * Np;
#pragma unroll
      for (int u = 0; u < ZU; ++u) v[q][u] = srow[min(lane + 64 * u, Np - 1)];
    }
  }
  float sv[PMAX / 8];
#pragma unroll
  for (int k = 0; k < PMAX / 8; ++k)
    sv[k] = sim_c[((size_t)g * P + min(slot + 8 * k, P - 1)) * Np + min(t0 + sp, Np - 1)];
  const float cnt_p = (float)cnt[g * PMAX + sp];
  const Box ob = load_box(box_patch_, g, Hp, Wp);
  const int nb = box_count(ob);
  __builtin_amdgcn_sched_barrier(0);
  if (t0 >= nb && assign_out == nullptr && tile != 0) return;   // (tile 0 always publishes the object's statistics)
  AS_STAMP(1, false);                            // box known
  const int ntiles = (nb + CS_TILE1 - 1) / CS_TILE1;
  AS_STAMP(2, true);                             // blind operand loads landed

  {                                             // maximum / previous-assignment density sums over the tiles
    float mx = -INFINITY, ds = 0.0f;
#pragma unroll
    for (int k = 0; k < PSU; ++k)
      if (slot + 8 * k < ntiles) { mx = fmaxf(mx, ps[k].x); ds += ps[k].y; }
    for (int t = slot + 8 * PSU; t < ntiles; t += 8) {
      const float* q = part_stats + (((size_t)g * nt1 + t) * PMAX + sp) * 2;
      mx = fmaxf(mx, q[0]);
      ds += q[1];
    }
    sh_mx[slot][sp] = mx; sh_ds[slot][sp] = ds;
  }
  __syncthreads();
  if (tid < PMAX) {
    const int p = tid;
    float maxsim = sh_mx[0][p], dens = sh_ds[0][p];
#pragma unroll
    for (int k = 1; k < 8; ++k) { maxsim = fmaxf(maxsim, sh_mx[k][p]); dens += sh_ds[k][p]; }
    if (nb < Np) maxsim = fmaxf(maxsim, 0.0f);          // out-of-box patches have cosine exactly 0
    float tau = tau0;
    if (it > 0) {                                       // update_density_batch (:882-908)
      const float c = p < P ? cnt_p : 0.0f;
      const float mean = c >= 1.0f ? dens / c : 0.0f;
      tau = fmaxf(1.0f - mean, 1e-10f);
      if (tau_out != nullptr && tile == 0 && p < P) tau_out[((size_t)(it - 1) * G + g) * P + p] = tau;
    }
    // iteration 0 divides by the python float temp*tau0 rounded ONCE to fp32 (stdroi:834 with scalar tau); later
    // iterations by the fp32 product temp * tau[g][p] of a tensor tau
    const float tt = it == 0 ? tt0 : temp * tau;
    st_s[p * 4 + 0] = tt; st_s[p * 4 + 1] = maxsim / tt; st_s[p * 4 + 3] = tau;
    zs_s[p * 2 + 0] = maxsim; zs_s[p * 2 + 1] = 1.4426950408889634f / tt;
  }
  __syncthreads();
  AS_STAMP(3, false);                            // tau / max per prototype
  // Z_p: wave w owns prototypes w, w+4, ... (<= 8 of them).  Every tile workgroup of the object repeats this sum,
  // so it uses the hardware exp2 on (sim - max) * log2(e)/(temp*tau): the difference is exact for the terms that
  // matter (close to the maximum), one rounding + v_exp_f32 instead of an IEEE division and a full expf per term.
  {
    float z[ZQ], mxs[ZQ], r2[ZQ];
#pragma unroll
    for (int q = 0; q < ZQ; ++q) {
      const int p = min(wave + 4 * q, P - 1);
      z[q] = 0.0f; mxs[q] = zs_s[p * 2 + 0]; r2[q] = zs_s[p * 2 + 1];
    }
    for (int tb = 0; tb < nb; tb += 64 * ZU) {
      if (tb > 0 || !eager) {                   // boxes beyond 768 patches: further trips
#pragma unroll
        for (int q = 0; q < ZQ; ++q) {
          const float* srow = sim_c + ((size_t)g * P + min(wave + 4 * q, P - 1)) * Np;
#pragma unroll
          for (int u = 0; u < ZU; ++u) v[q][u] = srow[min(tb + lane + 64 * u, nb - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < ZQ; ++q) {
        if (wave + 4 * q < P) {                  // wave-uniform
#pragma unroll
          for (int u = 0; u < ZU; ++u) {
            const float e = __builtin_amdgcn_exp2f((v[q][u] - mxs[q]) * r2[q]);
            z[q] += (tb + lane + 64 * u < nb) ? e : 0.0f;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < ZQ; ++q) {
      const int p = wave + 4 * q;
      const float zz = wave_sum(z[q]);
      if (lane == 0 && p < P) st_s[p * 4 + 2] = zz + (float)(Np - nb) * __builtin_amdgcn_exp2f((0.0f - mxs[q]) * r2[q]);
    }
  }
  __syncthreads();
  AS_STAMP(4, false);                            // Z per prototype
  if (tile == 0 && tid < P * 4) stats[(size_t)g * PMAX * 4 + tid] = st_s[tid];

  // user-visible assignment of out-of-box patches (never read back by the iteration itself)
  if (assign_out != nullptr) {
    const int n = tile * CS_TILE1 + tid;
    if (tid < CS_TILE1 && n < Np && !in_box(ob, n, Wp)) assign_out[(size_t)g * Np + n] = outside_cluster(st_s, P);
  }
  // this tile's 32 patches: thread (patch, slot) scores prototypes slot, slot+8, ..., then the 8 slots are merged
  // with the sequential scan's rule: highest weight, ties -> lowest prototype index
  {
    float bw = -INFINITY;
    int bp = PMAX;
#pragma unroll
    for (int k = 0; k < PMAX / 8; ++k) {
      const int p = slot + 8 * k;
      if (p < P) {
        const float w = expf(sv[k] / st_s[p * 4 + 0] - st_s[p * 4 + 1]) / st_s[p * 4 + 2];
        if (w > bw) { bw = w; bp = p; }
      }
    }
    __syncthreads();                               // sh_mx / sh_ds are free again
    sh_mx[slot][sp] = bw;
    sh_ds[slot][sp] = __int_as_float(bp);
  }
  __syncthreads();
  // the cluster every out-of-box patch falls into (the aggregation's member counts need it): one lane per prototype and a
  // shuffle argmax with the sequential scan's rule (strict >, so ties go to the lowest index) -- here the statistics are in
  // LDS; in the aggregation kernel the same scan was a dependent global round trip + 20 expf on the launch's critical path
  if (tile == 0 && wave == 1) {              // wave 0 merges the tile's assignment below; wave 1 is idle here
    const int p = lane & 31;
    float w = (lane < 32 && p < P) ? expf(0.0f / st_s[p * 4 + 0] - st_s[p * 4 + 1]) / st_s[p * 4 + 2] : -INFINITY;
    int bp = (lane < 32 && p < P) ? p : PMAX;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float w2 = __shfl_xor(w, o);
      const int p2 = __shfl_xor(bp, o);
      if (w2 > w || (w2 == w && p2 < bp)) { w = w2; bp = p2; }
    }
    if (lane == 0) oc_out[g] = bp < P ? bp : 0;
  }
  if (tid < CS_TILE1 && t0 + tid < nb) {
    const int t = t0 + tid;
    int best = 0;
    float bw = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = sh_mx[k][tid];
      const int p = __float_as_int(sh_ds[k][tid]);
      if (w > bw || (w == bw && p < best)) { bw = w; best = p; }
    }
    aw[(size_t)g * Np + t] = make_int2(best, __float_as_int(bw));
    if (assign_out != nullptr) assign_out[(size_t)g * Np + box_patch(ob, t, Wp)] = best;
  }
  AS_STAMP(5, false);                            // assignment stored (issued)
}

// ---- shift iteration, pass 3: cluster-wise aggregation, channel-major ---------------------------------------
// grid (C / 32, G).  A workgroup owns 32 channels of ALL prototypes of its object, so there are no cross-workgroup
// partials: thread (member slot ms, channel quad cq) walks the in-box patches ms, ms+32, ... in order and adds
// w * feat into ITS slot of the patch's cluster (LDS, one float4 per thread and prototype); the 32 member slots are
// then summed in a fixed order.  Also: squared-norm partial of the new prototypes (for the next similarity pass)
// and, from channel block 0, the member counts (for the density).
__global__ __launch_bounds__(CS_NT) void shift_aggregate_kernel(const float* __restrict__ feat,
                                                                const int2* __restrict__ aw,
                                                                const int32_t* __restrict__ oc_in,
                                                                const int32_t* __restrict__ box_patch_,
                                                                const int32_t* __restrict__ obj_img,
                                                                float* __restrict__ prot, float* __restrict__ pn2,
                                                                int32_t* __restrict__ cnt, int C, int Hp, int Wp,
                                                                int P, size_t fbs AS_STAMP_ARG) {
  __shared__ float4 acc_s[PMAX * CS_NT];
  __shared__ int cnt_s[PMAX];
  const int Np = Hp * Wp;
  const int chunk = blockIdx.x, nchunk = gridDim.x, g = blockIdx.y, tid = threadIdx.x;
  AS_STAMP(0, false);
  const Box ob = load_box(box_patch_, g, Hp, Wp);
  const int nb = box_count(ob), bw = max(box_w(ob), 1);
  const int b = obj_img[g];
  const int oc = oc_in[g];                      // cluster of the out-of-box patches (shift_assign), requested early
  const int cq = tid & 7, ms = tid >> 3;
  const bool count_here = chunk == 0 && cq == 0;
  if (tid < PMAX) cnt_s[tid] = 0;
  __syncthreads();
  AS_STAMP(1, false);                            // box known

  const float* fb = feat + (size_t)b * fbs + chunk * SH_CH + cq * 4;
  const int2* awg = aw + (size_t)g * Np;
  constexpr int U = 16;
  bool zeroed = false;
  for (int j0 = ms; j0 < nb || !zeroed; j0 += 32 * U) {
    float4 f[U];
    int2 m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = max(min(j0 + 32 * u, nb - 1), 0);
      const int ty = j / bw;
      const int n = min((ob.y0 + ty) * Wp + ob.x0 + (j - ty * bw), Np - 1);
      f[u] = *reinterpret_cast<const float4*>(fb + (size_t)n * C);
      m[u] = awg[j];
    }
    __builtin_amdgcn_sched_barrier(0);          // all 2*U loads issued before the first use
    if (j0 == ms) AS_STAMP(2, true);            // first batch of (feature, assignment) loads landed
    if (!zeroed) {                              // a thread only ever touches its own accumulator slots: no barrier
      for (int p = 0; p < P; ++p) acc_s[p * CS_NT + tid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      zeroed = true;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j0 + 32 * u < nb) {
        const float w = __int_as_float(m[u].y);
        float4 cur = acc_s[m[u].x * CS_NT + tid];
        cur.x = fmaf(w, f[u].x, cur.x); cur.y = fmaf(w, f[u].y, cur.y);
        cur.z = fmaf(w, f[u].z, cur.z); cur.w = fmaf(w, f[u].w, cur.w);
        acc_s[m[u].x * CS_NT + tid] = cur;
        if (count_here) atomicAdd(&cnt_s[m[u].x], 1);
      }
    }
  }
  __syncthreads();
  AS_STAMP(3, false);                            // member-slot accumulation done

  const float* accf = reinterpret_cast<const float*>(acc_s);
  for (int o = tid; o < PMAX * SH_CH; o += CS_NT) {       // uniform trip count: the shuffles below need whole waves
    const int p = o >> 5, c = o & 31;
    float v = 0.0f;
    if (p < P) {
      const float* base = accf + ((size_t)p * CS_NT + (c >> 2)) * 4 + (c & 3);
      float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        v0 += base[(k + 0) * 32]; v1 += base[(k + 1) * 32];
        v2 += base[(k + 2) * 32]; v3 += base[(k + 3) * 32];
      }
      v = (v0 + v1) + (v2 + v3);
      prot[((size_t)g * P + p) * C + chunk * SH_CH + c] = v;
    }
    float sq = v * v;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    if (c == 0 && p < P) pn2[((size_t)g * nchunk + chunk) * PMAX + p] = sq;
  }
  if (chunk == 0 && tid < P) {
    int c = cnt_s[tid];
    if (nb < Np && oc == tid) c += Np - nb;
    cnt[g * PMAX + tid] = c;
  }
  AS_STAMP(4, false);                            // prototypes / norms / counts stored (issued)
}

struct WsLayout {
  size_t pn2, cnt, oc, stats, part_stats, sim_c, aw, total;
  int nt1, nchunk;
};
WsLayout ws_layout(int B, int C, int Np, int G, int P) {
  WsLayout w;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  w.nt1 = as_ceil_div(Np, CS_TILE1);
  w.nchunk = as_ceil_div(C, SH_CH);
  size_t o = 0;
  w.pn2 = o; o = al(o + (size_t)G * w.nchunk * PMAX * 4);
  w.cnt = o; o = al(o + (size_t)G * PMAX * 4);
  w.oc = o; o = al(o + (size_t)G * 4);
  w.stats = o; o = al(o + (size_t)G * PMAX * 16);
  w.part_stats = o; o = al(o + (size_t)G * w.nt1 * PMAX * 8);
  w.sim_c = o; o = al(o + (size_t)G * P * Np * 4);
  w.aw = o; o = al(o + (size_t)G * Np * 8);
  w.total = o;
  return w;
}

}  // namespace

void as_shift_final_sim_launch(const float* feat, const float* prot, const int32_t* box_patch, const int32_t* obj_img,
                               const int2* aw, float* sim_out, float* part_stats, int B, int C, int Hp, int Wp, int P, int G,
                               int nt1, size_t fbs, hipStream_t s);           // shift_final.hip

#ifdef AS_SHIFT_STAMPS
extern "C" int as_shift_stamps_read(void* dst, size_t bytes, int clear) {
  if (bytes > sizeof(g_stamps)) bytes = sizeof(g_stamps);
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)) != hipSuccess || hipMemset(p, 0, sizeof(g_stamps)) != hipSuccess) return -2;
  }
  return 0;
}
#endif

extern "C" size_t as_cosine_shift_workspace_bytes(int B, int C, int Hp, int Wp, int G, int P) {
  if (B <= 0 || C <= 0 || Hp <= 0 || Wp <= 0 || G <= 0 || P <= 0) return 0;
  return ws_layout(B, C, Hp * Wp, G, P).total;
}

extern "C" int as_cosine_shift_strided(const float* feat, long long feat_batch_stride, const int32_t* box_patch,
                                       const int32_t* obj_img, const float* prot_in, float* prot_out, double tau0_d, double temp_d,
                                       int n_shift, float* sim_out, int32_t* assign_out, float* tau_out, void* ws, size_t ws_bytes,
                                       int B, int C, int Hp, int Wp, int G, int P, as_stream_t stream);

extern "C" int as_cosine_shift(const float* feat, const int32_t* box_patch, const int32_t* obj_img, const float* prot_in,
                               float* prot_out, double tau0_d, double temp_d, int n_shift, float* sim_out,
                               int32_t* assign_out, float* tau_out, void* ws, size_t ws_bytes, int B, int C, int Hp, int Wp,
                               int G, int P, as_stream_t stream) {
  return as_cosine_shift_strided(feat, (long long)Hp * Wp * C, box_patch, obj_img, prot_in, prot_out, tau0_d, temp_d, n_shift,
                                 sim_out, assign_out, tau_out, ws, ws_bytes, B, C, Hp, Wp, G, P, stream);
}

extern "C" int as_cosine_shift_strided(const float* feat, long long feat_batch_stride, const int32_t* box_patch,
                                       const int32_t* obj_img, const float* prot_in, float* prot_out, double tau0_d, double temp_d,
                                       int n_shift, float* sim_out, int32_t* assign_out, float* tau_out, void* ws, size_t ws_bytes,
                                       int B, int C, int Hp, int Wp, int G, int P, as_stream_t stream) {
  AS_REQUIRE(feat && box_patch && obj_img && prot_in && prot_out && sim_out && ws, AS_E_BADARG, "as_cosine_shift: null pointer");
  AS_REQUIRE(feat_batch_stride >= (long long)Hp * Wp * C && feat_batch_stride % 4 == 0, AS_E_BADARG,
             "as_cosine_shift: image stride %lld floats (>= Hp * Wp * C, multiple of 4: 16-byte rows)", feat_batch_stride);
  const size_t fbs = (size_t)feat_batch_stride;
  AS_REQUIRE(B > 0 && Hp > 0 && Wp > 0 && G > 0 && n_shift >= 0, AS_E_BADARG, "as_cosine_shift: bad sizes");
  AS_REQUIRE(P > 0 && P <= PMAX, AS_E_UNSUPPORTED, "as_cosine_shift: P=%d prototypes per object (max %d)", P, PMAX);
  AS_REQUIRE(C % SH_CH == 0 && C <= 4 * CS_NT, AS_E_UNSUPPORTED,
             "as_cosine_shift: C=%d must be a multiple of %d and <= 1024", C, SH_CH);
  AS_REQUIRE(G <= 256, AS_E_UNSUPPORTED, "as_cosine_shift: %d objects per call (max 256)", G);
  const int Np = Hp * Wp;
  const float tau0 = (float)tau0_d, temp = (float)temp_d, tt0 = (float)(temp_d * tau0_d);
  const WsLayout L = ws_layout(B, C, Np, G, P);
  AS_REQUIRE(ws_bytes >= L.total, AS_E_WORKSPACE, "as_cosine_shift: workspace %zu < %zu bytes", ws_bytes, L.total);
  AS_REQUIRE(L.nchunk <= 64, AS_E_UNSUPPORTED, "as_cosine_shift: C=%d gives more than 64 channel blocks", C);
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  float* pn2 = (float*)(w + L.pn2);
  int32_t* cnt = (int32_t*)(w + L.cnt);
  int32_t* oc = (int32_t*)(w + L.oc);
  float* stats = (float*)(w + L.stats);
  float* part_stats = (float*)(w + L.part_stats);
  float* sim_c = (float*)(w + L.sim_c);
  int2* aw = (int2*)(w + L.aw);

  if (n_shift == 0 && prot_out != prot_in)
    (void)hipMemcpyAsync(prot_out, prot_in, (size_t)G * P * C * 4, hipMemcpyDeviceToDevice, s);
  // three launches per iteration (a dependent launch boundary costs ~1.5 us, a grid barrier more):
  //   similarity tiles -> statistics + assignment per tile -> channel-major aggregation (no partial prototypes).
  // Iteration 0 reads the caller's seeds; every aggregation writes prot_out (which may alias prot_in).
  for (int it = 0; it < n_shift; ++it) {
    hipLaunchKernelGGL(shift_sim_kernel, dim3(L.nt1, G), dim3(S1_NT), 0, s, feat, it == 0 ? prot_in : prot_out,
                       it == 0 ? nullptr : pn2, L.nchunk, box_patch, obj_img, it > 0 ? aw : nullptr, sim_c, part_stats,
                       C, Hp, Wp, P, L.nt1, fbs AS_STAMP_VAL(it < 5 ? it * 3 : 15));
    int32_t* aout = assign_out ? assign_out + (size_t)it * G * Np : nullptr;
    hipLaunchKernelGGL(shift_assign_kernel, dim3(L.nt1, G), dim3(CS_NT), 0, s, sim_c, part_stats, cnt, box_patch,
                       stats, tau_out, aw, aout, oc, tau0, temp, tt0, it, Hp, Wp, P, G, L.nt1 AS_STAMP_VAL(it < 5 ? it * 3 + 1 : 15));
    hipLaunchKernelGGL(shift_aggregate_kernel, dim3(L.nchunk, G), dim3(CS_NT), 0, s, feat, aw, oc, box_patch,
                       obj_img, prot_out, pn2, cnt, C, Hp, Wp, P, fbs AS_STAMP_VAL(it < 5 ? it * 3 + 2 : 15));
  }
  // final similarity on the UNMASKED map (+ the density of the last assignment for tau_out)
  as_shift_final_sim_launch(feat, n_shift > 0 ? prot_out : prot_in, box_patch, obj_img, (n_shift > 0 && tau_out) ? aw : nullptr,
                            sim_out, part_stats, B, C, Hp, Wp, P, G, L.nt1, fbs, s);
  if (n_shift > 0 && tau_out != nullptr)
    hipLaunchKernelGGL(stats_kernel, dim3(P, G), dim3(CS_NT), 0, s, sim_out, part_stats, cnt, box_patch, stats,
                       tau_out, tau0, temp, tt0, n_shift, Hp, Wp, P, G, L.nt1, 1);
  AS_CHECK_LAUNCH("cosine_shift");
  return AS_OK;
}

// =====================================================================================================
// Cosine-affinity refinement on the patch grid (stdroi:668-707 get_refined_similarity), one image.
//
//   level 0      : cos(seed, feat) -> output map (box mask + keep-the-winner when is_select)
//   level l >= 1 : seeds <- similarity-weighted mean feature over the patches with cos >= tau * max;
//                  cos again; the map carried to the next level is box-masked (:696), level 0's is not (:672)
//
// Two launches per level: an aggregation owned per 32-channel block (no partials) and a similarity pass whose
// epilogue applies the masking / selection and reduces the row maxima the next level thresholds with.  The norms
// of both operands are accumulated from the MFMA operand fragments, so there is no separate norm pass.
// =====================================================================================================
namespace {

// grid (tiles).  work[g][n] = the map the next level consumes, out[g][n] = this level's output map,
// peak[g] (ordered-uint, zeroed by the caller) = max_n work[g][n].
__global__ __launch_bounds__(S1_NT) void refine_sim_kernel(const float* __restrict__ feat, const float* __restrict__ seeds,
                                                           const int32_t* __restrict__ boxes, float* __restrict__ work,
                                                           float* __restrict__ out, unsigned* __restrict__ peak, int G,
                                                           int Gp, int C, int Hp, int Wp, int is_select, int mask_work) {
  __shared__ float red[8][32][33];
  __shared__ float nrm[2][16][32];
  __shared__ float val[32][33];
  __shared__ Box sbox[32];
  const int Np = Hp * Wp;
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the boxes of the selection group, read once (the epilogue below used to load them per map inside its loops: two
  // dependent L2 round trips per map in the launch's critical path)
  if (tid < 32) sbox[tid] = (tid < is_select && tid < G) ? load_box(boxes, tid, Hp, Wp) : Box{0, 0, -1, -1};
  const int li = lane & 31, half = lane >> 5;
  const int n_mine = min(tile * CS_TILE1 + li, Np - 1);
  const float* frow = feat + (size_t)n_mine * C;
  const float* prow = seeds + (size_t)min(li, Gp - 1) * C;
  const float pkeep = li < Gp ? 1.0f : 0.0f;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float qa = 0.0f, qb = 0.0f;                    // squared-norm partials of this lane's operand fragments
  const int nsteps = (C + 15) / 16;
  constexpr int SU = 6;
  for (int s = wave; s < nsteps; s += 8 * SU) {
    Frag<float> fa[SU], fb[SU];
    float keep[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int su = s + 8 * u;
      const bool ok = su < nsteps && su * 16 + half * 8 + 8 <= C;
      const int k0 = ok ? su * 16 + half * 8 : 0;
      fb[u].load16B(frow + k0);
      fa[u].load16B(prow + k0);
      keep[u] = ok ? 1.0f : 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);           // all operand loads in flight before the first MFMA
#pragma unroll
    for (int u = 0; u < SU; ++u) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        fa[u].v[t] *= pkeep * keep[u];
        fb[u].v[t] *= keep[u];
        qa = fmaf(fa[u].v[t], fa[u].v[t], qa);
        qb = fmaf(fb[u].v[t], fb[u].v[t], qb);
      }
      acc = mma32(fa[u], fb[u], acc);            // D[g][n]
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][acc_row(r, half)][li] = acc[r];
  nrm[0][wave * 2 + half][li] = qa;
  nrm[1][wave * 2 + half][li] = qb;
  __syncthreads();

  {
    const int nn = tid & 31, pq = tid >> 5;
    float fn = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) fn += nrm[1][k][nn];
    const float fin = 1.0f / fmaxf(sqrtf(fn), COS_EPS);
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int p = pq + 16 * qd;
      float sn = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) sn += nrm[0][k][p];
      const float v = (((red[0][p][nn] + red[1][p][nn]) + (red[2][p][nn] + red[3][p][nn])) +
                       ((red[4][p][nn] + red[5][p][nn]) + (red[6][p][nn] + red[7][p][nn])));
      val[p][nn] = v * (1.0f / fmaxf(sqrtf(sn), COS_EPS)) * fin;
    }
  }
  __syncthreads();

  // epilogue: one thread per (map, patch) -- masking, the selection group's winner, both outputs, and the row maximum of
  // the map over the tile's 32 patches (a half-wave reduction, one atomic per map and workgroup).  The first `is_select`
  // maps form the selection group (0 = none, the refinement of the background seeds; Gp = all, the foreground seeds; in
  // between = both seed sets refined by one call, the group first).  (Round 5: this ran as a loop over the maps on 32
  // lanes -- 24 serial iterations of stores + shuffles + an atomic for the part-similarity call.)
  const int nsel = is_select;
  for (int idx = tid; idx < 32 * Gp; idx += S1_NT) {
    const int g = idx >> 5, pl = idx & 31;
    const int n = tile * CS_TILE1 + pl;
    const bool nvalid = n < Np;
    const int nc = min(n, Np - 1);
    auto masked = [&](int gg) {
      float v = val[gg][pl];
      if (gg < nsel && gg < G) v = v * (in_box(sbox[gg], nc, Wp) ? 1.0f : 0.0f);
      return v;
    };
    const float raw = val[g][pl];
    float o = raw, wv = raw;
    if (g < nsel) {
      int best = 0;
      float bv = -INFINITY;
      for (int gg = 0; gg < nsel; ++gg) {
        const float v = masked(gg);
        if (v > bv) { bv = v; best = gg; }       // strict: ties keep the lowest map index
      }
      const float mv = masked(g);
      o = g == best ? mv : 0.0f;
      wv = mask_work ? mv : raw;
    }
    if (nvalid) {
      out[(size_t)g * Np + n] = o;
      work[(size_t)g * Np + n] = wv;
    }
    float m = nvalid ? wv : -INFINITY;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (pl == 0) atomicMax(&peak[g], f2ord(m));
  }
}

// grid (C / 32, Gp).  seeds_out[g][c] = sum_n w[n] feat[n][c] / clamp(sum_n w[n], 1e-8) with w = work[g] zeroed
// below tau * peak[g] (:687-691).  Each wave compacts its quarter of the patch range (ballot + popcount, patch
// order), then thread (member slot, channel quad) accumulates the surviving patches slot, slot+32, ...
__global__ __launch_bounds__(CS_NT) void refine_aggregate_kernel(const float* __restrict__ feat,
                                                                 const float* __restrict__ work,
                                                                 const unsigned* __restrict__ peak_in,
                                                                 unsigned* __restrict__ peak_zero, float tau,
                                                                 float* __restrict__ seeds_out, int C, int Np, int Q) {
  extern __shared__ unsigned char dyn_s[];
  int* n_s = reinterpret_cast<int*>(dyn_s);                 // [4][Q]
  float* w_s = reinterpret_cast<float*>(dyn_s) + 4 * Q;     // [4][Q]
  __shared__ int cnt_s[4];
  __shared__ float wpart_s[CS_NT];
  __shared__ float4 part_s[CS_NT];
  const int chunk = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (chunk == 0 && g == 0 && tid < PMAX) peak_zero[tid] = 0u;
  const float thr = ord2f(peak_in[g]) * tau;
  const float* wrow = work + (size_t)g * Np;

  {                                              // wave-local compaction of patches [wave*Q, wave*Q + Q)
    int cnt = 0;
    constexpr int R = 8;
    const int lo = wave * Q, hi = min(lo + Q, Np);
    for (int r0 = lo; r0 < hi; r0 += 64 * R) {
      float v[R];
#pragma unroll
      for (int k = 0; k < R; ++k) v[k] = wrow[min(r0 + 64 * k + lane, Np - 1)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int n = r0 + 64 * k + lane;
        float w = v[k];
        if (w < thr) w = 0.0f;                   // cos_map1[cos_map1 < thr] *= 0
        const bool live = n < hi && w != 0.0f;
        const unsigned long long m = __ballot(live);
        if (live) {
          const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
          n_s[wave * Q + pos] = n;
          w_s[wave * Q + pos] = w;
        }
        cnt += __popcll(m);
      }
    }
    if (lane == 0) cnt_s[wave] = cnt;
  }
  __syncthreads();
  const int c0 = cnt_s[0], c1 = c0 + cnt_s[1], c2 = c1 + cnt_s[2], nlive = c2 + cnt_s[3];
  auto slot_of = [&](int i) {                     // i-th surviving patch in patch order -> index into n_s / w_s
    return i < c0 ? i : i < c1 ? Q + (i - c0) : i < c2 ? 2 * Q + (i - c1) : 3 * Q + (i - c2);
  };

  float wp = 0.0f;
  for (int i = tid; i < nlive; i += CS_NT) wp += w_s[slot_of(i)];
  wpart_s[tid] = wp;

  const int cq = tid & 7, ms = tid >> 3;
  const float* fb = feat + chunk * SH_CH + cq * 4;
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  constexpr int U = 8;
  for (int i0 = ms; i0 < nlive; i0 += 32 * U) {
    float4 f[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sl = slot_of(min(i0 + 32 * u, nlive - 1));
      f[u] = *reinterpret_cast<const float4*>(fb + (size_t)n_s[sl] * C);
      w[u] = i0 + 32 * u < nlive ? w_s[sl] : 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc.x = fmaf(w[u], f[u].x, acc.x); acc.y = fmaf(w[u], f[u].y, acc.y);
      acc.z = fmaf(w[u], f[u].z, acc.z); acc.w = fmaf(w[u], f[u].w, acc.w);
    }
  }
  part_s[tid] = acc;
  __syncthreads();
  for (int o = CS_NT / 2; o > 0; o >>= 1) {      // total weight, fixed tree
    if (tid < o) wpart_s[tid] += wpart_s[tid + o];
    __syncthreads();
  }
  if (tid < SH_CH) {
    const float* pf = reinterpret_cast<const float*>(part_s) + (tid >> 2) * 4 + (tid & 3);
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      v0 += pf[(k + 0) * 32]; v1 += pf[(k + 1) * 32]; v2 += pf[(k + 2) * 32]; v3 += pf[(k + 3) * 32];
    }
    seeds_out[(size_t)g * C + chunk * SH_CH + tid] = ((v0 + v1) + (v2 + v3)) / fmaxf(wpart_s[0], 1e-8f);
  }
}

struct RefineWs { size_t work, peak, total; int nt1, Q; };
RefineWs refine_ws(int C, int Np, int Gp) {
  RefineWs w;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  w.nt1 = as_ceil_div(Np, CS_TILE1);
  w.Q = as_round_up(as_ceil_div(Np, 4), 64);
  size_t o = 0;
  w.work = o; o = al(o + (size_t)Gp * Np * 4);
  w.peak = o; o = al(o + (size_t)2 * PMAX * 4);
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
  AS_REQUIRE(is_select >= 0 && is_select <= Gp && (is_select == 0 || G <= is_select), AS_E_BADARG,
             "as_refine_similarity: selection group of %d maps (0..Gp, at least the G boxed ones)", is_select);
  AS_REQUIRE(C % SH_CH == 0 && C <= 4 * CS_NT && Hp > 0 && Wp > 0 && refine_times >= 0, AS_E_UNSUPPORTED,
             "as_refine_similarity: C=%d must be a multiple of %d and <= 1024", C, SH_CH);
  const int Np = Hp * Wp;
  const RefineWs L = refine_ws(C, Np, Gp);
  AS_REQUIRE(ws_bytes >= L.total, AS_E_WORKSPACE, "as_refine_similarity: workspace %zu < %zu bytes", ws_bytes, L.total);
  const size_t lds = (size_t)8 * 4 * L.Q;
  AS_REQUIRE(lds <= 120 * 1024, AS_E_UNSUPPORTED, "as_refine_similarity: %d patches exceed the aggregation list", Np);
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)ws;
  float* work = (float*)(w + L.work);
  unsigned* peak = (unsigned*)(w + L.peak);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)refine_aggregate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);

  // (the per-map peaks feed the aggregation of the NEXT level only: without refinement levels nothing reads them)
  if (refine_times > 0) (void)hipMemsetAsync(peak, 0, (size_t)2 * PMAX * 4, s);
  const float* cur = seeds;
  for (int lvl = 0; lvl <= refine_times; ++lvl) {
    if (lvl > 0) {
      hipLaunchKernelGGL(refine_aggregate_kernel, dim3(C / SH_CH, Gp), dim3(CS_NT), lds, s, feat, work,
                         peak + ((lvl - 1) & 1) * PMAX, peak + (lvl & 1) * PMAX, tau, seeds_out, C, Np, L.Q);
      cur = seeds_out;
    }
    hipLaunchKernelGGL(refine_sim_kernel, dim3(L.nt1), dim3(S1_NT), 0, s, feat, cur, boxes, work,
                       maps + (size_t)lvl * Gp * Np, peak + (lvl & 1) * PMAX, G, Gp, C, Hp, Wp, is_select,
                       lvl > 0 ? 1 : 0);
  }
  if (refine_times == 0 && seeds_out != seeds)
    (void)hipMemcpyAsync(seeds_out, seeds, (size_t)Gp * C * 4, hipMemcpyDeviceToDevice, s);
  AS_CHECK_LAUNCH("refine_similarity");
  return AS_OK;
}
