// Full-resolution instance maps from the patch-grid cosine refinement, gfx950.
//
// Restates the tail of get_cosine_similarity_refined_map (reference
// mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py:1010-1019) with normalize_map (:1037-1040)
// and decouple_instance (:1042-1046) for all refinement levels and objects of an image at once:
//   up_fg, up_bg = bilinear x16 of the patch-grid maps        ret = (1 - up_bg) * up_fg
//   map_fg = ret / clamp(max ret, 1e-8)
//   nb = up_bg / (max up_bg + 1e-8); nf = ret / (max ret + 1e-8); bg = nb + (1 - (nf*0.5 + nb*0.5))
//   map_bg = bg / clamp(max bg, 1e-8)
// The reference materialises five full-resolution temporaries per map; here the upsampled values are
// recomputed from the (L2-resident) 64x64 maps in each of three passes, so HBM traffic is the two
// output tensors only.  Compiled with -ffp-contract=off so every step rounds like the ATen ops.
#include "bilinear.h"

namespace {

constexpr int RF_NT = 256;

struct MapMeta { unsigned max_ret, max_upbg, max_bg, pad; };

__global__ void meta_init_kernel(MapMeta* meta, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  MapMeta m;
  m.max_ret = 0u; m.max_upbg = 0u; m.max_bg = 0u; m.pad = 0u;   // 0 encodes "below every float"
  meta[i] = m;
}

__device__ __forceinline__ float block_max(float v, float* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = RF_NT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}

// PASS 1: max ret, max up_bg      PASS 2: max bg      PASS 3: write map_fg, map_bg
template <int PASS>
__global__ __launch_bounds__(RF_NT) void instance_maps_kernel(const float* __restrict__ sim_fg,
                                                              const float* __restrict__ sim_bg,
                                                              MapMeta* __restrict__ meta, float* __restrict__ map_fg,
                                                              float* __restrict__ map_bg, int G, int Gp, int Hp, int Wp,
                                                              int up) {
  __shared__ float sh[RF_NT];
  const int lg = blockIdx.y;                   // l * G + g
  const int l = lg / G, g = lg - l * G;
  const int H = Hp * up, W = Wp * up, Np = Hp * Wp;
  const float* fg = sim_fg + ((size_t)l * Gp + g) * Np;
  const float* bg = sim_bg + ((size_t)l * G + g) * Np;
  const float sy = (float)Hp / (float)H, sx = (float)Wp / (float)W;
  float mret = 0.0f, mupbg = 0.0f, mbg = 0.0f;
  float a1 = -INFINITY, a2 = -INFINITY;
  if (PASS >= 2) { mret = ord2f(meta[lg].max_ret); mupbg = ord2f(meta[lg].max_upbg); }
  if (PASS == 3) mbg = ord2f(meta[lg].max_bg);
  for (int i = blockIdx.x * RF_NT + threadIdx.x; i < H * W; i += gridDim.x * RF_NT) {
    const int y = i / W, x = i - y * W;
    const Lerp ly = lerp_axis(y, Hp, sy), lx = lerp_axis(x, Wp, sx);
    const float ufg = bilerp(fg, Wp, ly, lx), ubg = bilerp(bg, Wp, ly, lx);
    const float ret = (1.0f - ubg) * ufg;
    if (PASS == 1) {
      a1 = fmaxf(a1, ret);
      a2 = fmaxf(a2, ubg);
    } else {
      const float nb = ubg / (mupbg + 1e-8f);
      const float nf = ret / (mret + 1e-8f);
      const float b = nb + (1.0f - (nf * 0.5f + nb * 0.5f));
      if (PASS == 2) {
        a1 = fmaxf(a1, b);
      } else {
        map_fg[(size_t)lg * H * W + i] = ret / fmaxf(mret, 1e-8f);
        map_bg[(size_t)lg * H * W + i] = b / fmaxf(mbg, 1e-8f);
      }
    }
  }
  if (PASS == 1) {
    const float r1 = block_max(a1, sh), r2 = block_max(a2, sh);
    if (threadIdx.x == 0) { atomicMax(&meta[lg].max_ret, f2ord(r1)); atomicMax(&meta[lg].max_upbg, f2ord(r2)); }
  } else if (PASS == 2) {
    const float r1 = block_max(a1, sh);
    if (threadIdx.x == 0) atomicMax(&meta[lg].max_bg, f2ord(r1));
  }
}

}  // namespace

extern "C" size_t as_instance_maps_workspace_bytes(int L, int G) {
  if (L <= 0 || G <= 0) return 0;
  return ((size_t)L * G * sizeof(MapMeta) + 255) / 256 * 256;
}

extern "C" int as_instance_maps(const float* sim_fg, const float* sim_bg, int L, int G, int Gp, int Hp, int Wp, int up,
                                float* map_fg, float* map_bg, void* ws, size_t ws_bytes, as_stream_t stream) {
  AS_REQUIRE(sim_fg && sim_bg && map_fg && map_bg && ws, AS_E_BADARG, "as_instance_maps: null pointer");
  AS_REQUIRE(L > 0 && G > 0 && Gp >= G && Hp > 0 && Wp > 0 && up > 0, AS_E_BADARG, "as_instance_maps: bad sizes");
  AS_REQUIRE(ws_bytes >= as_instance_maps_workspace_bytes(L, G), AS_E_WORKSPACE, "as_instance_maps: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  MapMeta* meta = (MapMeta*)ws;
  const int n = L * G;
  const size_t hw = (size_t)Hp * up * Wp * up;
  const int bx = (int)((hw + RF_NT * 4 - 1) / (RF_NT * 4));
  hipLaunchKernelGGL(meta_init_kernel, dim3(as_ceil_div(n, 64)), dim3(64), 0, s, meta, n);
  hipLaunchKernelGGL((instance_maps_kernel<1>), dim3(bx, n), dim3(RF_NT), 0, s, sim_fg, sim_bg, meta, map_fg, map_bg, G,
                     Gp, Hp, Wp, up);
  hipLaunchKernelGGL((instance_maps_kernel<2>), dim3(bx, n), dim3(RF_NT), 0, s, sim_fg, sim_bg, meta, map_fg, map_bg, G,
                     Gp, Hp, Wp, up);
  hipLaunchKernelGGL((instance_maps_kernel<3>), dim3(bx, n), dim3(RF_NT), 0, s, sim_fg, sim_bg, meta, map_fg, map_bg, G,
                     Gp, Hp, Wp, up);
  AS_CHECK_LAUNCH("instance_maps");
  return AS_OK;
}

// =====================================================================================================
// Thresholded + eroded candidate masks inside per-map crops.
//   fg candidates of get_mask_points_single_box_cos_map_fg_bg (stdroi:442: erode(map > max*thr, 21) on the box
//   crop), its bg candidates (:443, no erosion) and the full-map erosion of get_semantic_centers (:2011).
// Erosion = min over the k x k window restricted to the crop (max_pool2d's implicit padding never wins),
// done as two 1-D passes on a byte mask.
// =====================================================================================================
namespace {

struct Crop { int x0, y0, x1, y1; };
__device__ __forceinline__ Crop load_crop(const int32_t* crops, int m, int H, int W) {
  Crop c;
  if (crops == nullptr) { c.x0 = 0; c.y0 = 0; c.x1 = W; c.y1 = H; return c; }
  c.x0 = min(max(crops[m * 4 + 0], 0), W); c.y0 = min(max(crops[m * 4 + 1], 0), H);
  c.x1 = min(max(crops[m * 4 + 2], 0), W); c.y1 = min(max(crops[m * 4 + 3], 0), H);
  return c;
}

__global__ void crop_meta_init_kernel(unsigned* mx, int32_t* counts, int M) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < M) { mx[m] = 0u; counts[m] = 0; }
}

__global__ __launch_bounds__(RF_NT) void crop_max_kernel(const float* __restrict__ maps, const int32_t* __restrict__ crops,
                                                         unsigned* __restrict__ mx, int H, int W) {
  __shared__ float sh[RF_NT];
  const int m = blockIdx.y;
  const Crop c = load_crop(crops, m, H, W);
  const int cw = max(c.x1 - c.x0, 0), ch = max(c.y1 - c.y0, 0);
  float v = -INFINITY;
  for (int i = blockIdx.x * RF_NT + threadIdx.x; i < cw * ch; i += gridDim.x * RF_NT) {
    const int y = c.y0 + i / cw, x = c.x0 + i % cw;
    v = fmaxf(v, maps[((size_t)m * H + y) * W + x]);
  }
  const float r = block_max(v, sh);
  if (threadIdx.x == 0 && cw * ch > 0) atomicMax(&mx[m], f2ord(r));
}

// MODE 0: bin = in_crop && map > thr          (counts if FINAL)
// MODE 1: out = AND of in[y][x-r..x+r] within the crop
// MODE 2: out = AND of in[y-r..y+r][x] within the crop (counts)
template <int MODE, bool FINAL>
__global__ __launch_bounds__(RF_NT) void crop_mask_kernel(const float* __restrict__ maps, const uint8_t* __restrict__ in,
                                                          const int32_t* __restrict__ crops,
                                                          const unsigned* __restrict__ mx, float thr, int relative,
                                                          int r, uint8_t* __restrict__ out, int32_t* __restrict__ counts,
                                                          int H, int W) {
  __shared__ int shc[RF_NT];
  const int m = blockIdx.y;
  const Crop c = load_crop(crops, m, H, W);
  const size_t base = (size_t)m * H * W;
  float t = thr;
  if (MODE == 0 && relative) t = ord2f(mx[m]) * thr;        // map.max() * thr, fp32
  int cnt = 0;
  for (int i = blockIdx.x * RF_NT + threadIdx.x; i < H * W; i += gridDim.x * RF_NT) {
    const int y = i / W, x = i - y * W;
    const bool inside = x >= c.x0 && x < c.x1 && y >= c.y0 && y < c.y1;
    bool v = false;
    if (inside) {
      if (MODE == 0) {
        v = maps[base + i] > t;
      } else if (MODE == 1) {
        v = true;
        for (int xx = max(x - r, c.x0); xx <= min(x + r, c.x1 - 1); ++xx) v = v && (in[base + (size_t)y * W + xx] != 0);
      } else {
        v = true;
        for (int yy = max(y - r, c.y0); yy <= min(y + r, c.y1 - 1); ++yy) v = v && (in[base + (size_t)yy * W + x] != 0);
      }
    }
    out[base + i] = v ? 1 : 0;
    cnt += v ? 1 : 0;
  }
  if (FINAL) {
    shc[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = RF_NT / 2; o > 0; o >>= 1) {
      if (threadIdx.x < o) shc[threadIdx.x] += shc[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0 && shc[0] > 0) atomicAdd(&counts[m], shc[0]);
  }
}

}  // namespace

extern "C" size_t as_crop_threshold_erode_workspace_bytes(int M, int H, int W) {
  if (M <= 0 || H <= 0 || W <= 0) return 0;
  return 2 * (((size_t)M * H * W + 255) / 256 * 256) + ((size_t)M * 4 + 255) / 256 * 256;
}

extern "C" int as_crop_threshold_erode(const float* maps, const int32_t* crops, float thr, int relative, int k,
                                       uint8_t* mask, int32_t* counts, void* ws, size_t ws_bytes, int M, int H, int W,
                                       as_stream_t stream) {
  AS_REQUIRE(maps && mask && counts && ws, AS_E_BADARG, "as_crop_threshold_erode: null pointer");
  AS_REQUIRE(M > 0 && H > 0 && W > 0 && k >= 1 && (k & 1) == 1, AS_E_BADARG, "as_crop_threshold_erode: bad sizes (k odd)");
  AS_REQUIRE(ws_bytes >= as_crop_threshold_erode_workspace_bytes(M, H, W), AS_E_WORKSPACE,
             "as_crop_threshold_erode: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t plane = ((size_t)M * H * W + 255) / 256 * 256;
  uint8_t* t0 = (uint8_t*)ws;
  uint8_t* t1 = t0 + plane;
  unsigned* mx = (unsigned*)(t1 + plane);
  const int bx = (int)(((size_t)H * W + RF_NT * 4 - 1) / (RF_NT * 4));
  const int r = k / 2;
  hipLaunchKernelGGL(crop_meta_init_kernel, dim3(as_ceil_div(M, 64)), dim3(64), 0, s, mx, counts, M);
  if (relative) hipLaunchKernelGGL(crop_max_kernel, dim3(bx, M), dim3(RF_NT), 0, s, maps, crops, mx, H, W);
  if (k == 1) {
    hipLaunchKernelGGL((crop_mask_kernel<0, true>), dim3(bx, M), dim3(RF_NT), 0, s, maps, (const uint8_t*)nullptr, crops,
                       mx, thr, relative, 0, mask, counts, H, W);
  } else {
    hipLaunchKernelGGL((crop_mask_kernel<0, false>), dim3(bx, M), dim3(RF_NT), 0, s, maps, (const uint8_t*)nullptr, crops,
                       mx, thr, relative, 0, t0, counts, H, W);
    hipLaunchKernelGGL((crop_mask_kernel<1, false>), dim3(bx, M), dim3(RF_NT), 0, s, maps, t0, crops, mx, thr, relative, r,
                       t1, counts, H, W);
    hipLaunchKernelGGL((crop_mask_kernel<2, true>), dim3(bx, M), dim3(RF_NT), 0, s, maps, t1, crops, mx, thr, relative, r,
                       mask, counts, H, W);
  }
  AS_CHECK_LAUNCH("crop_threshold_erode");
  return AS_OK;
}
