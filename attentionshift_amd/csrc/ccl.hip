// Connected components (8-connectivity) and the fused CAM -> box stage for gfx950.
//
//   as_ccl_2d    replaces cc_torch.connected_components_labeling (reference
//                mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py:23,68; the upstream CUDA
//                source is absent from the reference tree).  label = 1 + min raster index of the
//                component (our numbering; the consumer only uses the partition, :69-86).
//   as_cam_boxes replaces, for all Lc*G maps of an image at once, the bilinear x16 upsample (:2279) and
//                get_bbox_from_cam_fast (:60-116): min-max normalise, threshold, CCL, area filter,
//                tight box, 'expand' about the point.
//
// as_ccl_2d, run-based union-find over a label image (blobs are huge; per-pixel unions would serialise on the
// root's atomics):
//   rowscan   one workgroup per image row: prefix-max scan of the last background column, so every
//             foreground pixel points at the first pixel of its horizontal run (no atomics)
//   merge     one union per (run, upper run) contact: a pixel links to the row above only where its own
//             run or the upper run starts (plus the two diagonal contacts)
//   compress  run starts find their root; then every pixel takes root = L[L[p]]
// Links always go from the larger raster index to the smaller (atomicMin), so a set's root is its minimum
// index whatever the scheduling: labels are bit-exact and deterministic.
// as_cam_boxes needs only the PARTITION (areas, kept set, extents) and never builds the label image: see the
// run-based path below (cam_minmax / cam_runs / cam_cc) and the fused sampling masks (cam_sample_masks).
// Compiled with -ffp-contract=off (bilinear.h).
#include "bilinear.h"

namespace {

constexpr int CC_NT = 256;

__device__ __forceinline__ int uf_find(const int32_t* Lc, int a) {
  int32_t* L = const_cast<int32_t*>(Lc);
  int p = __hip_atomic_load(&L[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != a) {
    a = p;
    p = __hip_atomic_load(&L[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return a;
}
__device__ __forceinline__ void uf_union(int32_t* L, int a, int b) {
  bool done = false;
  while (!done) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a < b) {
      const int old = atomicMin(&L[b], a);
      done = (old == b);
      b = old;
    } else if (b < a) {
      const int old = atomicMin(&L[a], b);
      done = (old == a);
      a = old;
    } else {
      done = true;
    }
  }
}

// ---- foreground predicates ------------------------------------------------------------------------
struct FgImage {            // plain binary image
  const uint8_t* img;
  __device__ __forceinline__ bool operator()(int m, int y, int x, int H, int W) const {
    return img[((size_t)m * H + y) * W + x] != 0;
  }
};
struct CamMeta {            // per map, in workspace
  unsigned mn, mx;          // ordered-uint encoded min / max of the upsampled map
  int max_area;
  int pad;
};
// ---- rowscan: grid (H, M).  L[p] = first pixel of p's run, or -1 -----------------------------------
template <typename Fg>
__global__ __launch_bounds__(CC_NT) void ccl_rowscan_kernel(Fg fg, int32_t* __restrict__ Lall, int H, int W) {
  __shared__ int wave_max[4];
  __shared__ int carry_s;
  const int y = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t* Lrow = Lall + ((size_t)m * H + y) * W;
  if (tid == 0) carry_s = -1;                     // column of the last background pixel so far
  __syncthreads();
  for (int x0 = 0; x0 < W; x0 += CC_NT) {
    const int x = x0 + tid;
    const bool isfg = x < W && fg(m, y, x, H, W);
    int v = (x < W && !isfg) ? x : -1;            // inclusive prefix max of background columns
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (lane >= o) v = max(v, u);
    }
    if (lane == 63) wave_max[wave] = v;
    __syncthreads();
    int before = carry_s;
    for (int w = 0; w < wave; ++w) before = max(before, wave_max[w]);
    const int lastbg = max(v, before);
    if (x < W) Lrow[x] = isfg ? (y * W + lastbg + 1) : -1;
    __syncthreads();
    if (tid == CC_NT - 1) carry_s = lastbg;
    __syncthreads();
  }
}

// ---- merge: one thread per pixel; unions only at run contacts ---------------------------------------
__global__ __launch_bounds__(CC_NT) void ccl_merge_kernel(int32_t* __restrict__ Lall, int M, int H, int W) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  const int HW = H * W;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW), p = (int)(i % HW);
  int32_t* L = Lall + (size_t)m * HW;
  if (L[p] < 0) return;
  const int y = p / W, x = p - y * W;
  if (y == 0) return;
  const int up = p - W;
  const bool n = L[up] >= 0;
  const bool w_fg = x > 0 && L[p - 1] >= 0;
  const bool nw = x > 0 && L[up - 1] >= 0;
  if (n) {
    if (!w_fg || !nw) uf_union(L, p, up);        // my run or the upper run starts at this column
  } else {
    const bool ne = x + 1 < W && L[up + 1] >= 0;
    const bool e_fg = x + 1 < W && L[p + 1] >= 0;
    if (nw && !w_fg) uf_union(L, p, up - 1);     // diagonal contact with a run ending at x-1
    if (ne && !e_fg) uf_union(L, p, up + 1);     // diagonal contact with a run starting at x+1
  }
}

// ---- compress: run starts -> root ---------------------------------------------------------------------
__global__ __launch_bounds__(CC_NT) void ccl_compress_runs_kernel(int32_t* __restrict__ Lall, int M, int H, int W) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  const int HW = H * W;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW), p = (int)(i % HW);
  int32_t* L = Lall + (size_t)m * HW;
  if (L[p] < 0) return;
  const int x = p % W;
  if (x > 0 && L[p - 1] >= 0) return;             // not a run start
  const int root = uf_find(L, p);
  __hip_atomic_store(&L[p], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every pixel: root = L[L[p]] (start pixels already hold their root and L[root] == root).
// every pixel takes its component's root (a run start already holds it; the others hold their run start)
__global__ __launch_bounds__(CC_NT) void ccl_finalize_kernel(int32_t* __restrict__ Lall, int M, int H, int W) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  const int HW = H * W;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW), p = (int)(i % HW);
  int32_t* L = Lall + (size_t)m * HW;
  const int s = L[p];
  const int x = p % W;
  const bool w_fg = x > 0 && L[p - 1] >= 0;        // read BEFORE anything is written
  if (s < 0 || !w_fg) return;
  // (a start entry is only ever rewritten with the same value, so concurrent in-place writes are benign)
  __hip_atomic_store(&L[p], __hip_atomic_load(&L[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(CC_NT) void ccl_plus1_kernel(int32_t* __restrict__ L, size_t total) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  if (i < total) L[i] = L[i] + 1;
}

// ------------------------------------------------------------------------------------------------
// CAM stage
// ------------------------------------------------------------------------------------------------
__global__ void cam_meta_init_kernel(CamMeta* meta, int M) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  CamMeta c;
  c.mn = 0xffffffffu; c.mx = 0u; c.max_area = 0; c.pad = 0;
  meta[m] = c;
}

// (q / den >= thr) without the division except in a 1e-6-wide relative band around the threshold: outside the band the
// sign of q - thr*den decides, and the IEEE quotient, being monotone and 6e-8-accurate, rounds to the same side.
struct ThrBand { float lo, hi, den, thr; };
__device__ __forceinline__ ThrBand thr_band(float thr, float den) {
  ThrBand b;
  const float t = thr * den, mg = fabsf(t) * 1e-6f + 1e-30f;
  b.lo = t - mg; b.hi = t + mg; b.den = den; b.thr = thr;
  return b;
}
__device__ __forceinline__ bool ge_thr(float q, const ThrBand& b) {
  if (q >= b.hi) return true;
  if (!(q > b.lo)) return false;                            // below the band (or NaN: the quotient compares false too)
  return q / b.den >= b.thr;
}

constexpr int CAM_RB = 16;        // rows per workgroup (minmax / sampling masks) or per wave (runs)

// min / max of the upsampled map (and optionally the map itself); grid (ceil(H / CAM_RB), M)
__global__ __launch_bounds__(CC_NT) void cam_minmax_kernel(const float* __restrict__ cams, CamMeta* __restrict__ meta,
                                                           float* __restrict__ cams_up, int Hp, int Wp, int up) {
  __shared__ float smn[CC_NT], smx[CC_NT];
  const int m = blockIdx.y, H = Hp * up, W = Wp * up;
  const float* src = cams + (size_t)m * Hp * Wp;
  const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
  const int y0 = blockIdx.x * CAM_RB, y1 = min(y0 + CAM_RB, H);
  float mn = INFINITY, mx = -INFINITY;
  for (int x = threadIdx.x; x < W; x += CC_NT) {
    ColLerp c;
    c.lx = lerp_axis(x, Wp, sx);
    int cur = -1;
    for (int y = y0; y < y1; ++y) {
      const Lerp ly = lerp_axis(y, Hp, sy);
      if (ly.i0 != cur) { col_refresh(c, src, Wp, ly); cur = ly.i0; }
      const float v = col_value(c, ly);
      if (cams_up != nullptr) cams_up[((size_t)m * H + y) * W + x] = v;
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
  }
  smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
  __syncthreads();
  for (int o = CC_NT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + o]);
      smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMin(&meta[m].mn, f2ord(smn[0]));
    atomicMax(&meta[m].mx, f2ord(smx[0]));
  }
}

// ---- run-based CAM -> box path (no per-pixel label array) ---------------------------------------------------
// The foreground of an upsampled CAM is a handful of blobs: a row holds a few runs.  The box stage only needs the
// PARTITION of the foreground (areas, kept set, extents; stdroi:69-96), so the components are built over runs:
//   cam_runs   one wave per image row: foreground bits by ballot, run (x0,x1) pairs written per row
//   cam_cc     one workgroup per map: union-find over the runs (8-connectivity: runs of adjacent rows whose column
//              ranges touch or overlap diagonally), areas, area filter, extents, the 'expand' box
// Links go from the larger run id to the smaller (atomicMin) and areas / extents are integers, so the result does
// not depend on scheduling; it is the same partition the per-pixel path labels (tested against it).
// one wave per RUNS_RB consecutive rows; grid (ceil(H / (4 * RUNS_RB)), M).  Lane l owns the columns 64 * s + l.
constexpr int RUNS_RB = 1;
constexpr int CAM_XS = 16;        // 64-column words per row held in registers (rows up to 1024 wide; wider: reloaded)
__global__ __launch_bounds__(CC_NT) void cam_runs_kernel(const float* __restrict__ cams, const CamMeta* __restrict__ meta,
                                                         float thr, uint32_t* __restrict__ runs,
                                                         int32_t* __restrict__ nruns, int Hp, int Wp, int up, int rmax) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.y, H = Hp * up, W = Wp * up;
  const int yb = (blockIdx.x * 4 + wave) * RUNS_RB;
  if (yb >= H) return;
  const float* src = cams + (size_t)m * Hp * Wp;
  const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
  const float mn = ord2f(meta[m].mn), den = fmaxf(ord2f(meta[m].mx) - mn, 1e-6f);       // stdroi:63-66
  const ThrBand band = thr_band(thr, den);                  // (v - mn) / den >= thr
  const int nwords = (W + 63) / 64;
  for (int w0 = 0; w0 < nwords; w0 += CAM_XS) {            // one trip for W <= 1024
    ColLerp c[CAM_XS];
#pragma unroll
    for (int s = 0; s < CAM_XS; ++s) c[s].lx = lerp_axis(min((w0 + s) * 64 + lane, W - 1), Wp, sx);
    int cur = -1;
    for (int y = yb; y < min(yb + RUNS_RB, H); ++y) {
      const Lerp ly = lerp_axis(y, Hp, sy);
      if (ly.i0 != cur) {
#pragma unroll
        for (int s = 0; s < CAM_XS; ++s) col_refresh(c[s], src, Wp, ly);
        cur = ly.i0;
      }
      uint32_t* out = runs + ((size_t)m * H + y) * rmax;
      // wave-uniform run state; for rows wider than CAM_XS words it is carried through nruns / the last run word
      int n = 0, open = -1;
      if (w0 > 0) {                                          // written by this wave's lane 0 a trip ago: bypass L1
        n = __hip_atomic_load(&nruns[(size_t)m * H + y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n > 0 && n <= rmax) {
          const uint32_t last = __hip_atomic_load(&out[n - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((int)(last >> 16) == w0 * 64 - 1) { open = (int)(last & 0xffffu); --n; }
        }
      }
#pragma unroll
      for (int s = 0; s < CAM_XS; ++s) {
        const int x0 = (w0 + s) * 64, x = x0 + lane;
        const bool isfg = x < W && ge_thr(col_value(c[s], ly) - mn, band);      // words beyond W: all background
        const unsigned long long b = __ballot(isfg);
        const unsigned long long prev = (b << 1) | (open >= 0 ? 1ull : 0ull);
        unsigned long long starts = b & ~prev;              // pixel is foreground, its left neighbour is not
        unsigned long long ends = ~b & prev;                // pixel is background, its left neighbour was foreground
        while (starts | ends) {
          const int is = starts ? __ffsll((long long)starts) - 1 : 64;
          const int ie = ends ? __ffsll((long long)ends) - 1 : 64;
          if (ie < is) {
            if (lane == 0 && n < rmax) out[n] = (uint32_t)open | ((uint32_t)(x0 + ie - 1) << 16);
            ++n; open = -1; ends &= ends - 1;
          } else {
            open = x0 + is; starts &= starts - 1;
          }
        }
      }
      if (open >= 0) {                                      // closes at the end of this trip's columns
        const int xe = min((w0 + CAM_XS) * 64, W) - 1;
        if (lane == 0 && n < rmax) out[n] = (uint32_t)open | ((uint32_t)xe << 16);
        ++n;
      }
      if (lane == 0) nruns[(size_t)m * H + y] = n;          // n > rmax: overflow, reported through status
    }
  }
}

constexpr int CCM_NT = 1024;
constexpr int CCM_HMAX = 2048;          // rows whose run offsets fit the LDS scan
constexpr int CCM_RCAP = 18432;         // runs per map whose parents + run words fit LDS (else the workspace arrays)

struct CcShared { int32_t* offs; int *s_max, *s_changed, *s_x0, *s_x1, *s_y0, *s_y1, *s_cnt; };

// Components over the runs of one map: hook-and-shortcut rounds (every adjacency hooks the larger of the two current
// parents onto the smaller, then every run jumps to its grandparent twice) until nothing changes.  A blob as tall as
// the image is a chain of hundreds of stacked runs: a find-based union walks such chains one dependent access at a
// time (55 us measured), the rounds need about log2(chain) passes of a few LDS operations per thread.  Parents only
// ever decrease and end at the component's smallest run id, whatever the scheduling.  `parent` / `rw` (run words,
// compact ids) are in LDS or, for maps with more runs than fit, in the workspace.
__device__ __forceinline__ void cam_cc_body(int32_t* parent, const uint32_t* rw, int32_t* __restrict__ area,
                                            const CcShared sh, int R, int H, float area_ratio) {
  const int tid = threadIdx.x;
  const int32_t* offs = sh.offs;
  int &s_max = *sh.s_max, &s_changed = *sh.s_changed, &s_x0 = *sh.s_x0, &s_x1 = *sh.s_x1, &s_y0 = *sh.s_y0,
      &s_y1 = *sh.s_y1, &s_cnt = *sh.s_cnt;
  for (int i = tid; i < R; i += CCM_NT) { parent[i] = i; area[i] = 0; }
  __syncthreads();
  for (;;) {
    if (tid == 0) s_changed = 0;
    __syncthreads();
    bool ch = false;
    for (int y = 1 + tid; y < H; y += CCM_NT) {            // hooks across the (y-1, y) row boundary
      const int a0 = offs[y], a1 = offs[y + 1], b0 = offs[y - 1], b1 = a0;
      int kb = b0;
      for (int ka = a0; ka < a1; ++ka) {
        const uint32_t ra = rw[ka];
        const int ax0 = (int)(ra & 0xffffu), ax1 = (int)(ra >> 16);
        while (kb < b1 && (int)(rw[kb] >> 16) + 1 < ax0) ++kb;          // upper runs entirely to the left
        for (int k = kb; k < b1; ++k) {
          if ((int)(rw[k] & 0xffffu) > ax1 + 1) break;                  // 8-connectivity: ranges touch diagonally
          const int pu = __hip_atomic_load(&parent[ka], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int pv = __hip_atomic_load(&parent[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pu != pv) { atomicMin(&parent[max(pu, pv)], min(pu, pv)); ch = true; }
        }
      }
    }
    if (ch) s_changed = 1;
    __syncthreads();
    const bool again = s_changed != 0;
    for (int rep = 0; rep < 2; ++rep) {                    // shortcut twice
      for (int i = tid; i < R; i += CCM_NT) {
        const int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(&parent[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
    if (!again) break;
  }
  // after a round without hooks every run's parent is its root (the last shortcuts flattened what the previous
  // round hooked); areas per root, largest area, then extents of the runs whose component passes the filter
  for (int i = tid; i < R; i += CCM_NT) {
    const uint32_t r = rw[i];
    atomicAdd(&area[parent[i]], (int)(r >> 16) - (int)(r & 0xffffu) + 1);
  }
  __syncthreads();
  for (int i = tid; i < R; i += CCM_NT)
    if (parent[i] == i) atomicMax(&s_max, __hip_atomic_load(&area[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __syncthreads();
  const float need = area_ratio * (float)s_max;            // stdroi:84: fp32 compare of an int area
  int x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1, cnt = 0;
  for (int y = tid; y < H; y += CCM_NT) {
    for (int k = offs[y]; k < offs[y + 1]; ++k) {
      const uint32_t r = rw[k];
      if ((float)__hip_atomic_load(&area[parent[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) {
        const int rx0 = (int)(r & 0xffffu), rx1 = (int)(r >> 16);
        x0 = min(x0, rx0); x1 = max(x1, rx1); y0 = min(y0, y); y1 = max(y1, y); cnt += rx1 - rx0 + 1;
      }
    }
  }
  if (cnt > 0) {
    atomicMin(&s_x0, x0); atomicMax(&s_x1, x1); atomicMin(&s_y0, y0); atomicMax(&s_y1, y1); atomicAdd(&s_cnt, cnt);
  }
  __syncthreads();
}

__global__ __launch_bounds__(CCM_NT) void cam_cc_kernel(const uint32_t* __restrict__ runs_all,
                                                        const int32_t* __restrict__ nruns_all,
                                                        int32_t* __restrict__ parent_all, int32_t* __restrict__ area_all,
                                                        uint32_t* __restrict__ rw_all,
                                                        const CamMeta* __restrict__ meta, const float* __restrict__ points,
                                                        float* __restrict__ boxes, int32_t* __restrict__ status,
                                                        float* __restrict__ minmax, float area_ratio, int H, int W,
                                                        int rmax) {
  __shared__ int32_t offs[CCM_HMAX + 1];
  __shared__ int32_t lds_parent[CCM_RCAP];
  __shared__ uint32_t lds_rw[CCM_RCAP];
  __shared__ int wsum[CCM_NT / 64];
  __shared__ int s_max, s_over, s_x0, s_x1, s_y0, s_y1, s_cnt, s_changed, s_base;
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* runs = runs_all + (size_t)m * H * rmax;
  const int32_t* nruns = nruns_all + (size_t)m * H;
  if (tid == 0) { s_max = 0; s_over = 0; s_x0 = 0x7fffffff; s_x1 = -1; s_y0 = 0x7fffffff; s_y1 = -1; s_cnt = 0; s_base = 0; }
  __syncthreads();
  // exclusive scan of the per-row run counts -> compact run ids offs[y] + k
  for (int y0 = 0; y0 < H; y0 += CCM_NT) {
    const int y = y0 + tid;
    int n = y < H ? nruns[y] : 0;
    if (n > rmax) { s_over = 1; n = rmax; }
    int v = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (lane >= o) v += u;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (y < H) offs[y] = before + v - n;
    __syncthreads();
    if (tid == CCM_NT - 1) s_base = before + v;
    __syncthreads();
  }
  const int R = s_base;
  if (tid == 0) offs[H] = R;
  __syncthreads();
  const bool fits = R <= CCM_RCAP;
  uint32_t* rw_g = rw_all + (size_t)m * H * rmax;
  for (int y = tid; y < H; y += CCM_NT)                   // compact the run words
    for (int k = offs[y]; k < offs[y + 1]; ++k) {
      const uint32_t r = runs[y * rmax + (k - offs[y])];
      if (fits) lds_rw[k] = r; else rw_g[k] = r;
    }
  __syncthreads();
  // two instantiations of the same body, so that the LDS one compiles to ds_ instructions instead of flat accesses
  // through a generic pointer; areas are touched once per run and live in the workspace either way
  int32_t* area = area_all + (size_t)m * H * rmax;
  CcShared sh{offs, &s_max, &s_changed, &s_x0, &s_x1, &s_y0, &s_y1, &s_cnt};
  if (fits) cam_cc_body(lds_parent, lds_rw, area, sh, R, H, area_ratio);
  else cam_cc_body(parent_all + (size_t)m * H * rmax, rw_g, area, sh, R, H, area_ratio);
  if (tid != 0) return;
  if (status != nullptr) status[m] = s_over ? -1 : s_cnt;
  if (minmax != nullptr) { minmax[m * 2 + 0] = ord2f(meta[m].mn); minmax[m * 2 + 1] = ord2f(meta[m].mx); }
  float* bx = boxes + (size_t)m * 4;
  if (s_cnt == 0 || s_over) { bx[0] = 0.f; bx[1] = 0.f; bx[2] = 1.f; bx[3] = 1.f; return; }
  const float xc = points[m * 2 + 0], yc = points[m * 2 + 1];
  const float xmin = (float)s_x0, xmax = (float)s_x1, ymin = (float)s_y0, ymax = (float)s_y1;
  float gx0, gx1, gy0, gy1;
  if (fabsf(xc - xmin) > fabsf(xc - xmax)) {
    gx0 = xmin; gx1 = xc * 2.0f - gx0; gx1 = gx1 < (float)W ? gx1 : (float)W;
  } else {
    gx1 = xmax; gx0 = xc * 2.0f - gx1; gx0 = gx0 > 0.0f ? gx0 : 0.0f;
  }
  if (fabsf(yc - ymin) > fabsf(yc - ymax)) {
    gy0 = ymin; gy1 = yc * 2.0f - gy0; gy1 = gy1 < (float)H ? gy1 : (float)H;
  } else {
    gy1 = ymax; gy0 = yc * 2.0f - gy1; gy0 = gy0 > 0.0f ? gy0 : 0.0f;
  }
  bx[0] = gx0; bx[1] = gy0; bx[2] = gx1; bx[3] = gy1;
}

// ---- seed-sampling candidate masks straight from the low-resolution CAMs (stdroi:329-333, 343-371, 1003-1007) --
// For the G selected maps of an image: nm = (up16(cam) - min) / (max - min) recomputed on the fly (same bilinear
// code as the box stage, so the values are the ones the upsampled map would hold), and in ONE pass the three
// candidate masks of sample_point_grid -- background nm < thr_bg per map, foreground nm >= thr_fg per map, shared
// background mean_g(nm) < thr_bg -- with their candidate counts.  masks [2G+1][H*W] uint8, counts [2G+1].
// grid (ceil(H / CSM_RB)); thread t owns the 4 columns 4t .. 4t+3 (and 4t + 1024 k), see ColLerp above.
constexpr int CSM_G = 8;          // maps whose column state is kept in registers per pass
constexpr int CSM_RB = 4;         // rows per workgroup (one launch per image: 256 workgroups at 1024 rows)
__global__ __launch_bounds__(CC_NT) void cam_sample_masks_kernel(const float* __restrict__ cams,
                                                                 const int32_t* __restrict__ map_idx,
                                                                 const float* __restrict__ minmax, int G, int Hp, int Wp,
                                                                 int up, float thr_bg, float thr_fg,
                                                                 uint8_t* __restrict__ masks, int32_t* __restrict__ counts,
                                                                 float* __restrict__ sum_ws) {
  __shared__ int cnt_s[2 * 32 + 1];
  const int H = Hp * up, W = Wp * up, tid = threadIdx.x;
  const size_t HW = (size_t)H * W;
  const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
  const int y0 = blockIdx.x * CSM_RB, y1 = min(y0 + CSM_RB, H);
  for (int k = tid; k < 2 * G + 1; k += CC_NT) cnt_s[k] = 0;
  __syncthreads();
  int c_supp = 0;
  for (int xb = tid * 4; xb < W; xb += CC_NT * 4) {          // W % 4 == 0
    Lerp lx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lx[j] = lerp_axis(xb + j, Wp, sx);
    // maps in passes of CSM_G; the running sum over maps (for the shared-background mask) stays in registers when
    // G <= CSM_G and goes through sum_ws otherwise
    for (int g0 = 0; g0 < G; g0 += CSM_G) {
      const int gn = min(CSM_G, G - g0);
      ColLerp c[CSM_G][4];
      float lo[CSM_G], den[CSM_G];
      const float* src[CSM_G];
      int cb[CSM_G], cf[CSM_G];
#pragma unroll
      for (int g = 0; g < CSM_G; ++g) {
        const int mi = map_idx[g0 + min(g, gn - 1)];
        src[g] = cams + (size_t)mi * Hp * Wp;
        lo[g] = minmax[mi * 2 + 0]; den[g] = minmax[mi * 2 + 1] - lo[g];
        cb[g] = 0; cf[g] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) c[g][j].lx = lx[j];
      }
      int cur = -1;
      for (int y = y0; y < y1; ++y) {
        const Lerp ly = lerp_axis(y, Hp, sy);
        if (ly.i0 != cur) {
#pragma unroll
          for (int g = 0; g < CSM_G; ++g)
            if (g < gn) {
#pragma unroll
              for (int j = 0; j < 4; ++j) col_refresh(c[g][j], src[g], Wp, ly);
            }
          cur = ly.i0;
        }
        const size_t pix = (size_t)y * W + xb;
        float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (g0 > 0) {
          const float4 sv = *reinterpret_cast<const float4*>(sum_ws + pix);
          sum[0] = sv.x; sum[1] = sv.y; sum[2] = sv.z; sum[3] = sv.w;
        }
#pragma unroll
        for (int g = 0; g < CSM_G; ++g) {
          if (g < gn) {
            uchar4 bg, fg;
            unsigned char* pb = reinterpret_cast<unsigned char*>(&bg);
            unsigned char* pf = reinterpret_cast<unsigned char*>(&fg);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float nm = (col_value(c[g][j], ly) - lo[g]) / den[g];
              sum[j] += nm;
              pb[j] = nm < thr_bg ? 1 : 0;
              pf[j] = nm >= thr_fg ? 1 : 0;
              cb[g] += pb[j]; cf[g] += pf[j];
            }
            *reinterpret_cast<uchar4*>(masks + (size_t)(g0 + g) * HW + pix) = bg;
            *reinterpret_cast<uchar4*>(masks + (size_t)(G + g0 + g) * HW + pix) = fg;
          }
        }
        if (g0 + gn == G) {
          uchar4 sp;
          unsigned char* ps = reinterpret_cast<unsigned char*>(&sp);
#pragma unroll
          for (int j = 0; j < 4; ++j) { ps[j] = (sum[j] / (float)G) < thr_bg ? 1 : 0; c_supp += ps[j]; }
          *reinterpret_cast<uchar4*>(masks + (size_t)(2 * G) * HW + pix) = sp;
        } else {
          *reinterpret_cast<float4*>(sum_ws + pix) = make_float4(sum[0], sum[1], sum[2], sum[3]);
        }
      }
#pragma unroll
      for (int g = 0; g < CSM_G; ++g)
        if (g < gn) {
          if (cb[g]) atomicAdd(&cnt_s[g0 + g], cb[g]);
          if (cf[g]) atomicAdd(&cnt_s[G + g0 + g], cf[g]);
        }
    }
  }
  if (c_supp) atomicAdd(&cnt_s[2 * G], c_supp);
  __syncthreads();
  for (int k = tid; k < 2 * G + 1; k += CC_NT)
    if (counts && cnt_s[k]) atomicAdd(&counts[k], cnt_s[k]);
}

inline int blocks_for(size_t total) { return (int)((total + CC_NT - 1) / CC_NT); }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int as_ccl_2d(const uint8_t* img, int32_t* labels, int M, int H, int W, as_stream_t stream) {
  AS_REQUIRE(img && labels, AS_E_BADARG, "as_ccl_2d: null pointer");
  AS_REQUIRE(M > 0 && H > 0 && W > 0 && (size_t)H * W < 0x7fffffffu, AS_E_BADARG, "as_ccl_2d: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)M * H * W;
  FgImage fg{img};
  // `labels` doubles as the parent array
  hipLaunchKernelGGL((ccl_rowscan_kernel<FgImage>), dim3(H, M), dim3(CC_NT), 0, s, fg, labels, H, W);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels, M, H, W);
  hipLaunchKernelGGL(ccl_compress_runs_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels, M, H, W);
  hipLaunchKernelGGL(ccl_finalize_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels, M, H, W);
  hipLaunchKernelGGL(ccl_plus1_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels, total);
  AS_CHECK_LAUNCH("ccl_2d");
  return AS_OK;
}

namespace {
struct CamWs { size_t runs, nruns, parent, area, rw, meta, total; int rmax; };
CamWs cam_ws(int M, int Hp, int Wp, int up) {
  CamWs w;
  const size_t H = (size_t)Hp * up;
  // a row of the upsampled map is piecewise linear between the Wp source columns: at most one threshold crossing
  // per interval, i.e. <= Wp/2 + 1 runs; Wp + 2 leaves room for rounding wiggles (overflow is reported, not hidden)
  w.rmax = Wp + 2;
  size_t o = 0;
  w.runs = o; o = al256(o + (size_t)M * H * w.rmax * 4);
  w.nruns = o; o = al256(o + (size_t)M * H * 4);
  w.parent = o; o = al256(o + (size_t)M * H * w.rmax * 4);
  w.area = o; o = al256(o + (size_t)M * H * w.rmax * 4);
  w.rw = o; o = al256(o + (size_t)M * H * w.rmax * 4);
  w.meta = o; o = al256(o + (size_t)M * sizeof(CamMeta));
  w.total = o;
  return w;
}
}  // namespace

extern "C" size_t as_cam_boxes_workspace_bytes(int M, int Hp, int Wp, int up) {
  if (M <= 0 || Hp <= 0 || Wp <= 0 || up <= 0) return 0;
  return cam_ws(M, Hp, Wp, up).total;
}

extern "C" int as_cam_boxes(const float* cams, const float* points, float cam_thr, float area_ratio, int M, int Hp,
                            int Wp, int up, float* boxes, int32_t* status, float* cams_up, float* minmax, void* ws,
                            size_t ws_bytes, as_stream_t stream) {
  AS_REQUIRE(cams && points && boxes && ws, AS_E_BADARG, "as_cam_boxes: null pointer");
  AS_REQUIRE(M > 0 && Hp > 0 && Wp > 0 && up > 0, AS_E_BADARG, "as_cam_boxes: bad sizes");
  AS_REQUIRE((size_t)Wp * up <= 0xffffu && Hp * up <= CCM_HMAX, AS_E_UNSUPPORTED,
             "as_cam_boxes: upsampled size %dx%d exceeds %dx65535", Hp * up, Wp * up, CCM_HMAX);
  const CamWs L = cam_ws(M, Hp, Wp, up);
  AS_REQUIRE(ws_bytes >= L.total, AS_E_WORKSPACE, "as_cam_boxes: workspace %zu < %zu bytes", ws_bytes, L.total);
  hipStream_t s = (hipStream_t)stream;
  const int H = Hp * up, W = Wp * up;
  char* w = (char*)ws;
  uint32_t* runs = (uint32_t*)(w + L.runs);
  int32_t* nruns = (int32_t*)(w + L.nruns);
  int32_t* parent = (int32_t*)(w + L.parent);
  int32_t* area = (int32_t*)(w + L.area);
  uint32_t* rw = (uint32_t*)(w + L.rw);
  CamMeta* meta = (CamMeta*)(w + L.meta);
  hipLaunchKernelGGL(cam_meta_init_kernel, dim3(as_ceil_div(M, 64)), dim3(64), 0, s, meta, M);
  // two atomics per workgroup on meta[m] (same-address atomics serialise at ~10 ns each: 64 workgroups per map)
  hipLaunchKernelGGL(cam_minmax_kernel, dim3(as_ceil_div(H, CAM_RB), M), dim3(CC_NT), 0, s, cams, meta, cams_up, Hp, Wp,
                     up);
  hipLaunchKernelGGL(cam_runs_kernel, dim3(as_ceil_div(H, 4 * RUNS_RB), M), dim3(CC_NT), 0, s, cams, meta, cam_thr, runs,
                     nruns, Hp, Wp, up, L.rmax);
  hipLaunchKernelGGL(cam_cc_kernel, dim3(M), dim3(CCM_NT), 0, s, runs, nruns, parent, area, rw, meta, points, boxes,
                     status, minmax, area_ratio, H, W, L.rmax);
  AS_CHECK_LAUNCH("cam_boxes");
  return AS_OK;
}

extern "C" size_t as_cam_sample_masks_workspace_bytes(int G, int Hp, int Wp, int up) {
  if (G <= CSM_G || Hp <= 0 || Wp <= 0 || up <= 0) return 0;
  return (size_t)Hp * up * Wp * up * sizeof(float);        // running sum over maps between register passes
}

extern "C" int as_cam_sample_masks(const float* cams, const int32_t* map_idx, const float* minmax, int G, int Hp, int Wp,
                                   int up, float thr_bg, float thr_fg, uint8_t* masks, int32_t* counts, void* ws,
                                   size_t ws_bytes, as_stream_t stream) {
  AS_REQUIRE(cams && map_idx && minmax && masks, AS_E_BADARG, "as_cam_sample_masks: null pointer");
  AS_REQUIRE(G > 0 && G <= 32 && Hp > 0 && Wp > 0 && up > 0 && up % 4 == 0, AS_E_UNSUPPORTED,
             "as_cam_sample_masks: G=%d maps (max 32), scale %d must be a multiple of 4", G, up);
  const size_t need = as_cam_sample_masks_workspace_bytes(G, Hp, Wp, up);
  AS_REQUIRE(ws_bytes >= need && (ws || !need), AS_E_WORKSPACE, "as_cam_sample_masks: workspace %zu < %zu bytes",
             ws_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  if (counts) (void)hipMemsetAsync(counts, 0, (size_t)(2 * G + 1) * 4, s);   // (NULL: a caller that counts the rows itself)
  // one atomic per counter per workgroup (256 workgroups at 1024 rows)
  hipLaunchKernelGGL(cam_sample_masks_kernel, dim3(as_ceil_div(Hp * up, CSM_RB)), dim3(CC_NT), 0, s, cams, map_idx,
                     minmax, G, Hp, Wp, up, thr_bg, thr_fg, masks, counts, (float*)ws);
  AS_CHECK_LAUNCH("cam_sample_masks");
  return AS_OK;
}
