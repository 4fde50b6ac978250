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
// Union-find with atomicMin linking larger roots under smaller ones: the root of a set is always its
// minimum raster index, so the result is independent of scheduling (bit-exact labels).  Integer-only
// atomics everywhere (areas, extents) => deterministic.
// This file is compiled with -ffp-contract=off: the bilinear weights must round exactly like ATen's
// (src = scale*(dst+0.5)-0.5 as separate mul/sub) and the two interpolation FMAs are explicit.
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

// parent init from a binary image
__global__ __launch_bounds__(CC_NT) void ccl_init_kernel(const uint8_t* __restrict__ img, int32_t* __restrict__ L,
                                                         size_t total, int HW) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  if (i >= total) return;
  const int p = (int)(i % HW);
  L[i] = img[i] ? p : -1;
}

// link every foreground pixel with its already-visited 8-neighbours (W, NW, N, NE)
__global__ __launch_bounds__(CC_NT) void ccl_merge_kernel(int32_t* __restrict__ Lall, int M, int H, int W) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  const int HW = H * W;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW), p = (int)(i % HW);
  int32_t* L = Lall + (size_t)m * HW;
  if (L[p] < 0) return;
  const int y = p / W, x = p - y * W;
  const bool up = y > 0;
  const bool n_fg = up && L[p - W] >= 0;
  if (n_fg) {
    uf_union(L, p, p - W);               // N connects NW and NE transitively
  } else if (up) {
    if (x > 0 && L[p - W - 1] >= 0) uf_union(L, p, p - W - 1);
    if (x + 1 < W && L[p - W + 1] >= 0) uf_union(L, p, p - W + 1);
  }
  if (x > 0 && L[p - 1] >= 0) uf_union(L, p, p - 1);
}

// path compression in place: every foreground pixel points straight at its root.  Safe while other
// threads still traverse: a parent entry only ever changes from one ancestor to a closer-to-root one.
__global__ __launch_bounds__(CC_NT) void ccl_compress_kernel(int32_t* __restrict__ Lall, int M, int HW) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW), p = (int)(i % HW);
  int32_t* L = Lall + (size_t)m * HW;
  if (L[p] < 0) return;
  const int root = uf_find(L, p);
  __hip_atomic_store(&L[p], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// after compression: parent (= root) -> label root + 1, background 0
__global__ __launch_bounds__(CC_NT) void ccl_plus1_kernel(int32_t* __restrict__ L, size_t total) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  if (i >= total) return;
  L[i] = L[i] + 1;
}
__global__ __launch_bounds__(CC_NT) void ccl_area_kernel(const int32_t* __restrict__ L, int32_t* __restrict__ area,
                                                         int M, int HW) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW);
  const int root = L[i];
  if (root >= 0) atomicAdd(&area[(size_t)m * HW + root], 1);
}

// ------------------------------------------------------------------------------------------------
// bilinear upsample, align_corners = False, ATen-exact (see oracle upsample_bilinear_explicit)
// ------------------------------------------------------------------------------------------------
struct CamMeta {      // per map, in workspace
  unsigned mn, mx;    // ordered-uint encoded min / max of the upsampled map
  int max_area;
  int x0, y0, x1, y1; // extent of the kept pixels
  int kept;
};

__global__ void cam_meta_init_kernel(CamMeta* meta, int M) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  CamMeta c;
  c.mn = 0xffffffffu; c.mx = 0u; c.max_area = 0;
  c.x0 = 0x7fffffff; c.y0 = 0x7fffffff; c.x1 = -1; c.y1 = -1; c.kept = 0;
  meta[m] = c;
}

// pass 1: min / max of the upsampled map; grid (blocks, M)
__global__ __launch_bounds__(CC_NT) void cam_minmax_kernel(const float* __restrict__ cams, CamMeta* __restrict__ meta,
                                                           float* __restrict__ cams_up, int Hp, int Wp, int up) {
  __shared__ float smn[CC_NT], smx[CC_NT];
  const int m = blockIdx.y, H = Hp * up, W = Wp * up;
  const float* src = cams + (size_t)m * Hp * Wp;
  const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = blockIdx.x * CC_NT + threadIdx.x; i < H * W; i += gridDim.x * CC_NT) {
    const int y = i / W, x = i - y * W;
    const float v = bilerp(src, Wp, lerp_axis(y, Hp, sy), lerp_axis(x, Wp, sx));
    if (cams_up != nullptr) cams_up[(size_t)m * H * W + i] = v;
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

// pass 2: normalise, threshold, parent init
__global__ __launch_bounds__(CC_NT) void cam_binarise_kernel(const float* __restrict__ cams,
                                                             const CamMeta* __restrict__ meta,
                                                             int32_t* __restrict__ L, int32_t* __restrict__ area,
                                                             float cam_thr, int Hp, int Wp, int up) {
  const int m = blockIdx.y, H = Hp * up, W = Wp * up;
  const float* src = cams + (size_t)m * Hp * Wp;
  const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
  const float mn = ord2f(meta[m].mn), mx = ord2f(meta[m].mx);
  const float den = fmaxf(mx - mn, 1e-6f);
  for (int i = blockIdx.x * CC_NT + threadIdx.x; i < H * W; i += gridDim.x * CC_NT) {
    const int y = i / W, x = i - y * W;
    const float v = bilerp(src, Wp, lerp_axis(y, Hp, sy), lerp_axis(x, Wp, sx));
    const float nv = (v - mn) / den;
    L[(size_t)m * H * W + i] = (nv >= cam_thr) ? i : -1;
    area[(size_t)m * H * W + i] = 0;
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

__global__ __launch_bounds__(CC_NT) void cam_extent_kernel(const int32_t* __restrict__ Lall,
                                                           const int32_t* __restrict__ area,
                                                           CamMeta* __restrict__ meta, float area_ratio, int M, int H,
                                                           int W) {
  const size_t i = (size_t)blockIdx.x * CC_NT + threadIdx.x;
  const int HW = H * W;
  if (i >= (size_t)M * HW) return;
  const int m = (int)(i / HW), p = (int)(i % HW);
  const int root = Lall[i];                           // compressed: parent == root
  if (root < 0) return;
  // reference: areas >= area_ratio * max_area  (int64 tensor vs fp32 scalar tensor -> fp32 compare)
  if ((float)area[(size_t)m * HW + root] >= area_ratio * (float)meta[m].max_area) {
    const int y = p / W, x = p - y * W;
    atomicMin(&meta[m].x0, x); atomicMax(&meta[m].x1, x);
    atomicMin(&meta[m].y0, y); atomicMax(&meta[m].y1, y);
    atomicAdd(&meta[m].kept, 1);
  }
}

// stdroi:97-115, box_method == 'expand'
__global__ void cam_box_kernel(const CamMeta* __restrict__ meta, const float* __restrict__ points,
                               float* __restrict__ boxes, int32_t* __restrict__ status, int M, int H, int W) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const CamMeta c = meta[m];
  if (status != nullptr) status[m] = c.kept;
  float* bx = boxes + (size_t)m * 4;
  if (c.kept == 0) { bx[0] = 0.f; bx[1] = 0.f; bx[2] = 1.f; bx[3] = 1.f; return; }
  const float xc = points[m * 2 + 0], yc = points[m * 2 + 1];
  const float xmin = (float)c.x0, xmax = (float)c.x1, ymin = (float)c.y0, ymax = (float)c.y1;
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

}  // namespace

extern "C" int as_ccl_2d(const uint8_t* img, int32_t* labels, int M, int H, int W, as_stream_t stream) {
  AS_REQUIRE(img && labels, AS_E_BADARG, "as_ccl_2d: null pointer");
  AS_REQUIRE(M > 0 && H > 0 && W > 0 && (size_t)H * W < 0x7fffffffu, AS_E_BADARG, "as_ccl_2d: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)M * H * W;
  // `labels` doubles as the parent array: init -> merge -> compress to roots (in place) -> +1
  hipLaunchKernelGGL(ccl_init_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, img, labels, total, H * W);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels, M, H, W);
  hipLaunchKernelGGL(ccl_compress_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels, M, H * W);
  hipLaunchKernelGGL(ccl_plus1_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, labels, total);
  AS_CHECK_LAUNCH("ccl_2d");
  return AS_OK;
}

extern "C" size_t as_cam_boxes_workspace_bytes(int M, int Hp, int Wp, int up) {
  if (M <= 0 || Hp <= 0 || Wp <= 0 || up <= 0) return 0;
  const size_t hw = (size_t)Hp * up * Wp * up;
  return 2 * (size_t)M * hw * sizeof(int32_t) + (((size_t)M * sizeof(CamMeta)) + 255) / 256 * 256;
}

extern "C" int as_cam_boxes(const float* cams, const float* points, float cam_thr, float area_ratio, int M, int Hp,
                            int Wp, int up, float* boxes, int32_t* status, float* cams_up, void* ws, size_t ws_bytes,
                            as_stream_t stream) {
  AS_REQUIRE(cams && points && boxes && ws, AS_E_BADARG, "as_cam_boxes: null pointer");
  AS_REQUIRE(M > 0 && Hp > 0 && Wp > 0 && up > 0, AS_E_BADARG, "as_cam_boxes: bad sizes");
  AS_REQUIRE(ws_bytes >= as_cam_boxes_workspace_bytes(M, Hp, Wp, up), AS_E_WORKSPACE,
             "as_cam_boxes: workspace %zu < %zu bytes", ws_bytes, as_cam_boxes_workspace_bytes(M, Hp, Wp, up));
  hipStream_t s = (hipStream_t)stream;
  const int H = Hp * up, W = Wp * up;
  const size_t hw = (size_t)H * W, total = (size_t)M * hw;
  int32_t* L = (int32_t*)ws;
  int32_t* area = L + total;
  CamMeta* meta = (CamMeta*)(area + total);
  const int bx = (int)((hw + CC_NT * 4 - 1) / (CC_NT * 4));
  hipLaunchKernelGGL(cam_meta_init_kernel, dim3(as_ceil_div(M, 64)), dim3(64), 0, s, meta, M);
  hipLaunchKernelGGL(cam_minmax_kernel, dim3(bx, M), dim3(CC_NT), 0, s, cams, meta, cams_up, Hp, Wp, up);
  hipLaunchKernelGGL(cam_binarise_kernel, dim3(bx, M), dim3(CC_NT), 0, s, cams, meta, L, area, cam_thr, Hp, Wp, up);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, M, H, W);
  hipLaunchKernelGGL(ccl_compress_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, M, (int)hw);
  hipLaunchKernelGGL(ccl_area_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, area, M, (int)hw);
  hipLaunchKernelGGL(cam_maxarea_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, area, meta, M, (int)hw);
  hipLaunchKernelGGL(cam_extent_kernel, dim3(blocks_for(total)), dim3(CC_NT), 0, s, L, area, meta, area_ratio, M, H, W);
  hipLaunchKernelGGL(cam_box_kernel, dim3(as_ceil_div(M, 64)), dim3(64), 0, s, meta, points, boxes, status, M, H, W);
  AS_CHECK_LAUNCH("cam_boxes");
  return AS_OK;
}
