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
// Run-based union-find (the CAM foreground is a few huge blobs; per-pixel unions would serialise on the
// root's atomics):
//   rowscan   one workgroup per image row: prefix-max scan of the last background column, so every
//             foreground pixel points at the first pixel of its horizontal run (no atomics)
//   merge     one union per (run, upper run) contact: a pixel links to the row above only where its own
//             run or the upper run starts (plus the two diagonal contacts)
//   compress  run starts find their root; then every pixel takes root = L[L[p]]
// Links always go from the larger raster index to the smaller (atomicMin), so a set's root is its minimum
// index whatever the scheduling: labels are bit-exact and deterministic.  Areas and extents are
// accumulated per run / per row with integer arithmetic only.
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
struct FgCam {              // upsample + min-max normalise + threshold, recomputed on the fly
  const float* cams;
  const CamMeta* meta;
  float thr;
  int Hp, Wp;
  __device__ __forceinline__ bool operator()(int m, int y, int x, int H, int W) const {
    const float* src = cams + (size_t)m * Hp * Wp;
    const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
    const float v = bilerp(src, Wp, lerp_axis(y, Hp, sy), lerp_axis(x, Wp, sx));
    const float mn = ord2f(meta[m].mn), mx = ord2f(meta[m].mx);
    return (v - mn) / fmaxf(mx - mn, 1e-6f) >= thr;       // stdroi:63-66
  }
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
// PLUS1: write the final label root+1 / 0.  area != null: add each run's length at its root.
template <bool PLUS1>
__global__ __launch_bounds__(CC_NT) void ccl_finalize_kernel(int32_t* __restrict__ Lall, int32_t* __restrict__ area,
                                                             int M, int H, int W) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  const int HW = H * W;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW), p = (int)(i % HW);
  int32_t* L = Lall + (size_t)m * HW;
  const int s = L[p];
  const int x = p % W;
  // everything this thread needs from its neighbours is read BEFORE anything is written
  const bool w_fg = x > 0 && L[p - 1] >= 0;
  const bool e_fg = x + 1 < W && L[p + 1] >= 0;
  if (s < 0) return;
  const bool is_start = !w_fg;
  // a start pixel already holds its root; any other pixel holds its run start, whose entry is the root.
  // (A start entry is only ever rewritten with the same value, so concurrent in-place writes are benign.)
  const int root = is_start ? s : __hip_atomic_load(&L[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (area != nullptr && !e_fg) {
    const int xs = is_start ? x : (s % W);
    atomicAdd(&area[(size_t)m * HW + root], x - xs + 1);
  }
  if (!is_start) __hip_atomic_store(&L[p], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// parent array -> labels: root + 1 for foreground, 0 for background
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

// min / max of the upsampled map (and optionally the map itself); zeroes the area array; grid (blocks, M)
__global__ __launch_bounds__(CC_NT) void cam_minmax_kernel(const float* __restrict__ cams, CamMeta* __restrict__ meta,
                                                           float* __restrict__ cams_up, int32_t* __restrict__ area,
                                                           int Hp, int Wp, int up) {
  __shared__ float smn[CC_NT], smx[CC_NT];
  const int m = blockIdx.y, H = Hp * up, W = Wp * up;
  const float* src = cams + (size_t)m * Hp * Wp;
  const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = blockIdx.x * CC_NT + threadIdx.x; i < H * W; i += gridDim.x * CC_NT) {
    const int y = i / W, x = i - y * W;
    const float v = bilerp(src, Wp, lerp_axis(y, Hp, sy), lerp_axis(x, Wp, sx));
    if (cams_up != nullptr) cams_up[(size_t)m * H * W + i] = v;
    area[(size_t)m * H * W + i] = 0;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
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

__global__ __launch_bounds__(CC_NT) void cam_maxarea_kernel(const int32_t* __restrict__ L,
                                                            const int32_t* __restrict__ area,
                                                            CamMeta* __restrict__ meta, int M, int HW) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW), p = (int)(i % HW);
  if (L[i] == p) atomicMax(&meta[m].max_area, area[i]);      // roots only
}

// per row: extent and count of the pixels whose component passes the area filter; grid (H, M)
struct RowMeta { int x0, x1, cnt, pad; };
__global__ __launch_bounds__(CC_NT) void cam_row_extent_kernel(const int32_t* __restrict__ Lall,
                                                               const int32_t* __restrict__ area,
                                                               const CamMeta* __restrict__ meta,
                                                               RowMeta* __restrict__ rows, float area_ratio, int H, int W) {
  __shared__ int s0[CC_NT], s1[CC_NT], sc[CC_NT];
  const int y = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
  const size_t base = ((size_t)m * H + y) * W;
  const float need = area_ratio * (float)meta[m].max_area;   // stdroi:84: fp32 compare of an int area
  int x0 = 0x7fffffff, x1 = -1, cnt = 0;
  for (int x = tid; x < W; x += CC_NT) {
    const int root = Lall[base + x];
    if (root >= 0 && (float)area[(size_t)m * H * W + root] >= need) {
      x0 = min(x0, x); x1 = max(x1, x); ++cnt;
    }
  }
  s0[tid] = x0; s1[tid] = x1; sc[tid] = cnt;
  __syncthreads();
  for (int o = CC_NT / 2; o > 0; o >>= 1) {
    if (tid < o) { s0[tid] = min(s0[tid], s0[tid + o]); s1[tid] = max(s1[tid], s1[tid + o]); sc[tid] += sc[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { RowMeta r; r.x0 = s0[0]; r.x1 = s1[0]; r.cnt = sc[0]; r.pad = 0; rows[(size_t)m * H + y] = r; }
}

// per map: reduce the rows, then the 'expand' box of stdroi:97-115; grid (M)
__global__ __launch_bounds__(CC_NT) void cam_box_kernel(const RowMeta* __restrict__ rows, const CamMeta* __restrict__ meta,
                                                        const float* __restrict__ points, float* __restrict__ boxes,
                                                        int32_t* __restrict__ status, float* __restrict__ minmax, int H,
                                                        int W) {
  __shared__ int s0[CC_NT], s1[CC_NT], sy0[CC_NT], sy1[CC_NT], sc[CC_NT];
  const int m = blockIdx.x, tid = threadIdx.x;
  int x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1, cnt = 0;
  for (int y = tid; y < H; y += CC_NT) {
    const RowMeta r = rows[(size_t)m * H + y];
    if (r.cnt > 0) { x0 = min(x0, r.x0); x1 = max(x1, r.x1); y0 = min(y0, y); y1 = max(y1, y); cnt += r.cnt; }
  }
  s0[tid] = x0; s1[tid] = x1; sy0[tid] = y0; sy1[tid] = y1; sc[tid] = cnt;
  __syncthreads();
  for (int o = CC_NT / 2; o > 0; o >>= 1) {
    if (tid < o) {
      s0[tid] = min(s0[tid], s0[tid + o]); s1[tid] = max(s1[tid], s1[tid + o]);
      sy0[tid] = min(sy0[tid], sy0[tid + o]); sy1[tid] = max(sy1[tid], sy1[tid + o]); sc[tid] += sc[tid + o];
    }
    __syncthreads();
  }
  if (tid != 0) return;
  if (status != nullptr) status[m] = sc[0];
  if (minmax != nullptr) { minmax[m * 2 + 0] = ord2f(meta[m].mn); minmax[m * 2 + 1] = ord2f(meta[m].mx); }
  float* bx = boxes + (size_t)m * 4;
  if (sc[0] == 0) { bx[0] = 0.f; bx[1] = 0.f; bx[2] = 1.f; bx[3] = 1.f; return; }
  const float xc = points[m * 2 + 0], yc = points[m * 2 + 1];
  const float xmin = (float)s0[0], xmax = (float)s1[0], ymin = (float)sy0[0], ymax = (float)sy1[0];
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
  hipLaunchKernelGGL((ccl_finalize_kernel<true>), dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels,
                     (int32_t*)nullptr, M, H, W);
  hipLaunchKernelGGL(ccl_plus1_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels, total);
  AS_CHECK_LAUNCH("ccl_2d");
  return AS_OK;
}

extern "C" size_t as_cam_boxes_workspace_bytes(int M, int Hp, int Wp, int up) {
  if (M <= 0 || Hp <= 0 || Wp <= 0 || up <= 0) return 0;
  const size_t hw = (size_t)Hp * up * Wp * up;
  return 2 * al256((size_t)M * hw * sizeof(int32_t)) + al256((size_t)M * sizeof(CamMeta)) +
         al256((size_t)M * Hp * up * sizeof(RowMeta));
}

extern "C" int as_cam_boxes(const float* cams, const float* points, float cam_thr, float area_ratio, int M, int Hp,
                            int Wp, int up, float* boxes, int32_t* status, float* cams_up, float* minmax, void* ws,
                            size_t ws_bytes, as_stream_t stream) {
  AS_REQUIRE(cams && points && boxes && ws, AS_E_BADARG, "as_cam_boxes: null pointer");
  AS_REQUIRE(M > 0 && Hp > 0 && Wp > 0 && up > 0, AS_E_BADARG, "as_cam_boxes: bad sizes");
  AS_REQUIRE(ws_bytes >= as_cam_boxes_workspace_bytes(M, Hp, Wp, up), AS_E_WORKSPACE,
             "as_cam_boxes: workspace %zu < %zu bytes", ws_bytes, as_cam_boxes_workspace_bytes(M, Hp, Wp, up));
  hipStream_t s = (hipStream_t)stream;
  const int H = Hp * up, W = Wp * up;
  const size_t hw = (size_t)H * W, total = (size_t)M * hw;
  char* w = (char*)ws;
  int32_t* L = (int32_t*)w;
  int32_t* area = (int32_t*)(w + al256(total * 4));
  CamMeta* meta = (CamMeta*)(w + 2 * al256(total * 4));
  RowMeta* rows = (RowMeta*)(w + 2 * al256(total * 4) + al256((size_t)M * sizeof(CamMeta)));
  const int bx = (int)((hw + CC_NT * 16 - 1) / (CC_NT * 16));
  hipLaunchKernelGGL(cam_meta_init_kernel, dim3(as_ceil_div(M, 64)), dim3(64), 0, s, meta, M);
  // two atomics per workgroup on meta[m]: grid-stride over at most 32 workgroups per map (same-address atomics
  // serialise at ~10 ns each; 256 workgroups x 42 maps made this pass atomic-bound)
  hipLaunchKernelGGL(cam_minmax_kernel, dim3(bx < 32 ? bx : 32, M), dim3(CC_NT), 0, s, cams, meta, cams_up, area, Hp, Wp, up);
  FgCam fg{cams, meta, cam_thr, Hp, Wp};
  hipLaunchKernelGGL((ccl_rowscan_kernel<FgCam>), dim3(H, M), dim3(CC_NT), 0, s, fg, L, H, W);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, M, H, W);
  hipLaunchKernelGGL(ccl_compress_runs_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, M, H, W);
  hipLaunchKernelGGL((ccl_finalize_kernel<false>), dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, area, M, H, W);
  hipLaunchKernelGGL(cam_maxarea_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, area, meta, M, (int)hw);
  hipLaunchKernelGGL(cam_row_extent_kernel, dim3(H, M), dim3(CC_NT), 0, s, L, area, meta, rows, area_ratio, H, W);
  hipLaunchKernelGGL(cam_box_kernel, dim3(M), dim3(CC_NT), 0, s, rows, meta, points, boxes, status, minmax, H, W);
  AS_CHECK_LAUNCH("cam_boxes");
  return AS_OK;
}
