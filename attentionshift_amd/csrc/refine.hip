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
// grid (ceil(H / IM_RB), L*G): a workgroup owns a block of rows, a thread its columns (x = tid, tid + 256, ...); the
// horizontal interpolation of a column is kept while the rows share a source row pair (bilinear.h, ColLerp).
constexpr int IM_RB = 4;          // rows per workgroup: 256 workgroups per map at 1024 rows (3 maps per image: 16 rows left most CUs idle)
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
  const int y0 = blockIdx.x * IM_RB, y1 = min(y0 + IM_RB, H);
  float mret = 0.0f, mupbg = 0.0f, mbg = 0.0f;
  float a1 = -INFINITY, a2 = -INFINITY;
  if (PASS >= 2) { mret = ord2f(meta[lg].max_ret); mupbg = ord2f(meta[lg].max_upbg); }
  if (PASS == 3) mbg = ord2f(meta[lg].max_bg);
  for (int x = threadIdx.x; x < W; x += RF_NT) {
    ColLerp cf, cb;
    cf.lx = lerp_axis(x, Wp, sx);
    cb.lx = cf.lx;
    int cur = -1;
    for (int y = y0; y < y1; ++y) {
      const Lerp ly = lerp_axis(y, Hp, sy);
      if (ly.i0 != cur) { col_refresh(cf, fg, Wp, ly); col_refresh(cb, bg, Wp, ly); cur = ly.i0; }
      const float ufg = col_value(cf, ly), ubg = col_value(cb, ly);
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
          const size_t i = (size_t)lg * H * W + (size_t)y * W + x;
          map_fg[i] = ret / fmaxf(mret, 1e-8f);
          map_bg[i] = b / fmaxf(mbg, 1e-8f);
        }
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
  // the reduction passes end in one or two atomics per workgroup on a per-map word (same-address atomics serialise
  // at ~10 ns): 256 workgroups per map at 1024 rows
  const int by = as_ceil_div(Hp * up, IM_RB);
  hipLaunchKernelGGL(meta_init_kernel, dim3(as_ceil_div(n, 64)), dim3(64), 0, s, meta, n);
  hipLaunchKernelGGL((instance_maps_kernel<1>), dim3(by, n), dim3(RF_NT), 0, s, sim_fg, sim_bg, meta, map_fg, map_bg, G,
                     Gp, Hp, Wp, up);
  hipLaunchKernelGGL((instance_maps_kernel<2>), dim3(by, n), dim3(RF_NT), 0, s, sim_fg, sim_bg, meta, map_fg, map_bg, G,
                     Gp, Hp, Wp, up);
  hipLaunchKernelGGL((instance_maps_kernel<3>), dim3(by, n), dim3(RF_NT), 0, s, sim_fg, sim_bg, meta, map_fg, map_bg, G,
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
  // one atomicMax per workgroup on mx[m]: keep the workgroup count per map small (same-address atomics serialise)
  if (relative) hipLaunchKernelGGL(crop_max_kernel, dim3(bx < 64 ? bx : 64, M), dim3(RF_NT), 0, s, maps, crops, mx, H, W);
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

// =====================================================================================================
// Per-part statistics of get_center_coord_with_feat (stdroi:222-262) for M similarity maps [M, Hp*Wp] in one launch:
// peak, centroid of the pixels at the peak (mean of their integer coordinates), number of pixels > 0.9, the centre in
// image coordinates ((x, y) + 0.5) * stride and whether it lies inside the owner's box.  One workgroup per map; the
// coordinate sums are integers, so the result does not depend on the reduction order.
//   out_c [M,2] (x,y) float, out_yx [M,2] (y,x) int32 = trunc of the centroid, out_area [M] int32, out_inside [M] uint8
// =====================================================================================================
namespace {
__global__ __launch_bounds__(RF_NT) void part_stats_kernel(const float* __restrict__ maps, const float* __restrict__ rois,
                                                           const int32_t* __restrict__ owner, float stride,
                                                           float* __restrict__ out_c, int32_t* __restrict__ out_yx,
                                                           int32_t* __restrict__ out_area, uint8_t* __restrict__ out_inside,
                                                           int Hp, int Wp) {
  __shared__ float sh[RF_NT];
  __shared__ int shi[4][RF_NT];
  const int m = blockIdx.x, tid = threadIdx.x, Np = Hp * Wp;
  const float* src = maps + (size_t)m * Np;
  float mx = -INFINITY;
  for (int n = tid; n < Np; n += RF_NT) mx = fmaxf(mx, src[n]);
  const float peak = block_max(mx, sh);
  int cnt = 0, sy = 0, sx = 0, area = 0;
  for (int n = tid; n < Np; n += RF_NT) {
    const float v = src[n];
    if (v >= peak) { const int y = n / Wp; ++cnt; sy += y; sx += n - y * Wp; }
    if (v > 0.9f) ++area;
  }
  shi[0][tid] = cnt; shi[1][tid] = sy; shi[2][tid] = sx; shi[3][tid] = area;
  __syncthreads();
  for (int o = RF_NT / 2; o > 0; o >>= 1) {
    if (tid < o)
#pragma unroll
      for (int q = 0; q < 4; ++q) shi[q][tid] += shi[q][tid + o];
    __syncthreads();
  }
  if (tid != 0) return;
  const float c = (float)shi[0][0];
  const float cy = (float)shi[1][0] / c, cx = (float)shi[2][0] / c;        // mean of nonzero().float() rows
  const float px = (cx + 0.5f) * stride, py = (cy + 0.5f) * stride;
  out_c[m * 2 + 0] = px; out_c[m * 2 + 1] = py;
  out_yx[m * 2 + 0] = (int)cy; out_yx[m * 2 + 1] = (int)cx;
  out_area[m] = shi[3][0];
  const float* b = rois + (size_t)owner[m] * 4;
  out_inside[m] = (px >= b[0] && px <= b[2] && py >= b[1] && py <= b[3]) ? 1 : 0;
}
}  // namespace

extern "C" int as_part_stats(const float* maps, const float* rois, const int32_t* owner, float stride, float* out_c,
                             int32_t* out_yx, int32_t* out_area, uint8_t* out_inside, int M, int Hp, int Wp,
                             as_stream_t stream) {
  AS_REQUIRE(maps && rois && owner && out_c && out_yx && out_area && out_inside, AS_E_BADARG, "as_part_stats: null pointer");
  AS_REQUIRE(M > 0 && Hp > 0 && Wp > 0, AS_E_BADARG, "as_part_stats: bad sizes");
  hipLaunchKernelGGL(part_stats_kernel, dim3(M), dim3(RF_NT), 0, (hipStream_t)stream, maps, rois, owner, stride, out_c,
                     out_yx, out_area, out_inside, Hp, Wp);
  AS_CHECK_LAUNCH("part_stats");
  return AS_OK;
}

// =====================================================================================================
// Part selection of get_center_coord_with_feat (stdroi:222-262) for all objects without a host decision.  Per object g
// the P slots (the first ngroups[g] exist) are visited in stable descending-area order; a slot is taken when its centre
// lies inside the object's box and its position in that order is <= num_points (`if i > num_max_obj: break`).  The
// taken slots of object 0, 1, ... are compacted in visiting order: row r of the outputs holds the r-th taken slot's
// centre (twice: the reference returns a clone), the object's label (twice), the object index, and the backbone
// feature at the slot's integer centroid.  split[g] = slots taken by object g, split[G] = their total; rows >= total
// are zero.  One workgroup walks the objects (a prefix over g is needed), a second launch copies the feature rows.
// =====================================================================================================
namespace {
constexpr int PS_NT = 256;
__global__ __launch_bounds__(PS_NT) void part_select_kernel(const int32_t* __restrict__ area, const uint8_t* __restrict__ inside,
                                                            const int32_t* __restrict__ ngroups, const float* __restrict__ c,
                                                            const int64_t* __restrict__ labels, int G, int P, int num_points,
                                                            float* __restrict__ coords, float* __restrict__ coords_org,
                                                            int64_t* __restrict__ out_labels, int64_t* __restrict__ labels_org,
                                                            int64_t* __restrict__ corres, int32_t* __restrict__ sel_slot,
                                                            int32_t* __restrict__ split) {
  __shared__ int a_s[PS_NT], rank_s[PS_NT], chosen_s[PS_NT];
  __shared__ int base_s;
  const int p = threadIdx.x;
  if (p == 0) base_s = 0;
  for (int i = p; i < G * P; i += PS_NT) {            // rows that stay unused read as zeros
    coords[2 * i] = coords[2 * i + 1] = coords_org[2 * i] = coords_org[2 * i + 1] = 0.0f;
    out_labels[i] = labels_org[i] = corres[i] = 0;
    sel_slot[i] = -1;
  }
  __syncthreads();
  for (int g = 0; g < G; ++g) {
    const bool valid = p < P && p < ngroups[g];
    const int a = valid ? area[g * P + p] : -1;
    a_s[p] = p < P ? a : -2;
    __syncthreads();
    int rank = 0;
    for (int q = 0; q < P; ++q) rank += (a_s[q] > a || (a_s[q] == a && q < p)) ? 1 : 0;
    const bool chosen = valid && inside[g * P + p] != 0 && rank <= num_points;
    rank_s[p] = rank;
    chosen_s[p] = chosen ? 1 : 0;
    __syncthreads();
    int pos = 0, n = 0;
    for (int q = 0; q < P; ++q) {
      n += chosen_s[q];
      pos += (chosen_s[q] && rank_s[q] < rank) ? 1 : 0;
    }
    if (chosen) {
      const int r = base_s + pos, slot = g * P + p;
      const float x = c[2 * slot], y = c[2 * slot + 1];
      coords[2 * r] = x; coords[2 * r + 1] = y;
      coords_org[2 * r] = x; coords_org[2 * r + 1] = y;
      out_labels[r] = labels[g]; labels_org[r] = labels[g];
      corres[r] = g;
      sel_slot[r] = slot;
    }
    __syncthreads();
    if (p == 0) { split[g] = n; base_s += n; }
    __syncthreads();
  }
  if (p == 0) split[G] = base_s;
}

// feats[r, :] = feat_tok[yx[slot_r].y * Wp + yx[slot_r].x, :]  (vit_feat[:, cy, cx] of the token-major features)
__global__ __launch_bounds__(PS_NT) void part_feats_kernel(const int32_t* __restrict__ sel_slot, const int32_t* __restrict__ yx,
                                                           const float* __restrict__ feat_tok, float* __restrict__ feats,
                                                           int C, int Wp) {
  const int r = blockIdx.x, slot = sel_slot[r];
  float* dst = feats + (size_t)r * C;
  if (slot < 0) {
    for (int e = threadIdx.x; e < C; e += PS_NT) dst[e] = 0.0f;
    return;
  }
  const float* src = feat_tok + ((size_t)yx[2 * slot] * Wp + yx[2 * slot + 1]) * C;
  for (int e = threadIdx.x; e < C; e += PS_NT) dst[e] = src[e];
}
}  // namespace

extern "C" int as_part_select(const int32_t* area, const uint8_t* inside, const int32_t* ngroups, const float* c,
                              const int32_t* yx, const int64_t* labels, const float* feat_tok, int G, int P, int C, int Wp,
                              int num_points, float* coords, float* coords_org, int64_t* out_labels, int64_t* labels_org,
                              int64_t* corres, float* feats, int32_t* sel_slot, int32_t* split, as_stream_t stream) {
  AS_REQUIRE(area && inside && ngroups && c && yx && labels && feat_tok && coords && coords_org && out_labels && labels_org &&
                 corres && feats && sel_slot && split, AS_E_BADARG, "as_part_select: null pointer");
  AS_REQUIRE(G > 0 && P > 0 && P <= PS_NT && C > 0 && Wp > 0, AS_E_BADARG, "as_part_select: need G > 0 and 0 < P <= 256 (P=%d)", P);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(part_select_kernel, dim3(1), dim3(PS_NT), 0, s, area, inside, ngroups, c, labels, G, P, num_points, coords,
                     coords_org, out_labels, labels_org, corres, sel_slot, split);
  AS_CHECK_LAUNCH("part_select");
  hipLaunchKernelGGL(part_feats_kernel, dim3(G * P), dim3(PS_NT), 0, s, (const int32_t*)sel_slot, yx, feat_tok, feats, C, Wp);
  AS_CHECK_LAUNCH("part_feats");
  return AS_OK;
}

// =====================================================================================================
// filter_maps (stdroi:263-271) for all G*P shifted prototypes in one launch: the share of a prototype's > sim_thr
// support that lies on the object's patch-grid foreground, keep = share >= pos_thr.  fg_inter holds multiples of 1/4
// and the support is 0/1, so both sums are exact in fp32 whatever the order.
// =====================================================================================================
namespace {
__global__ __launch_bounds__(RF_NT) void filter_parts_kernel(const float* __restrict__ sim, const float* __restrict__ fg_inter,
                                                             float sim_thr, float pos_thr, uint8_t* __restrict__ keep, int P,
                                                             int Np) {
  __shared__ float sa[RF_NT], sb[RF_NT];
  const int gp = blockIdx.x, g = gp / P, tid = threadIdx.x;
  const float* srow = sim + (size_t)gp * Np;
  const float* frow = fg_inter + (size_t)g * Np;
  float a = 0.0f, b = 0.0f;
  for (int n = tid; n < Np; n += RF_NT) {
    const float sup = srow[n] > sim_thr ? 1.0f : 0.0f;
    a += frow[n] * sup;
    b += sup;
  }
  sa[tid] = a; sb[tid] = b;
  __syncthreads();
  for (int o = RF_NT / 2; o > 0; o >>= 1) {
    if (tid < o) { sa[tid] += sa[tid + o]; sb[tid] += sb[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) keep[gp] = (sa[0] / fmaxf(sb[0], 1e-6f)) >= pos_thr ? 1 : 0;
}

// The first K distinct values of M uniform draws floor(u * n) per object (= the head of a random permutation of the n
// candidates, stdroi:447), split into positive / negative candidate ranks.  flag |= an object with n < 4K candidates
// or fewer than K distinct draws (the caller then takes the host path for that image).
__global__ void draw_distinct_kernel(const int32_t* __restrict__ counts /*(n_pos, n_neg) of object g at g*csg + {0, csk}*/,
                                     int csg, int csk, const float* __restrict__ u /*[G,M]*/, int32_t* __restrict__ rank_pos,
                                     int32_t* __restrict__ rank_neg, uint8_t* __restrict__ is_pos,
                                     int32_t* __restrict__ flag, int G, int M, int K) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const int n_pos = counts[g * csg], n = n_pos + counts[g * csg + csk];
  int got = 0;
  int picked[32];
  // the first 32 draws in one batch of independent loads (the scan below would otherwise pay one memory round trip per draw)
  float ub[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) ub[j] = u[g * M + min(j, M - 1)];
  for (int j = 0; j < M && got < K; ++j) {
    int c = (int)((j < 32 ? ub[j] : u[g * M + j]) * (float)n);
    c = min(c, max(n - 1, 0));
    bool dup = false;
    for (int q = 0; q < got; ++q) dup = dup || picked[q] == c;
    if (!dup) picked[got++] = c;
  }
  if (n < 4 * K || got < K) atomicOr(flag, 1);
  for (int q = 0; q < K; ++q) {
    const int r = q < got ? picked[q] : 0;
    const bool pos = r < n_pos;
    is_pos[g * K + q] = pos ? 1 : 0;
    rank_pos[g * K + q] = pos ? r : 0;
    rank_neg[g * K + q] = pos ? 0 : r - n_pos;
  }
}
}  // namespace

extern "C" int as_filter_parts(const float* sim, const float* fg_inter, float sim_thr, float pos_thr, uint8_t* keep, int G,
                               int P, int Np, as_stream_t stream) {
  AS_REQUIRE(sim && fg_inter && keep, AS_E_BADARG, "as_filter_parts: null pointer");
  AS_REQUIRE(G > 0 && P > 0 && Np > 0, AS_E_BADARG, "as_filter_parts: bad sizes");
  hipLaunchKernelGGL(filter_parts_kernel, dim3(G * P), dim3(RF_NT), 0, (hipStream_t)stream, sim, fg_inter, sim_thr, pos_thr,
                     keep, P, Np);
  AS_CHECK_LAUNCH("filter_parts");
  return AS_OK;
}

extern "C" int as_draw_distinct(const int32_t* counts, int count_stride_g, int count_stride_k, const float* u, int32_t* rank_pos,
                                int32_t* rank_neg, uint8_t* is_pos, int32_t* flag, int G, int M, int K, as_stream_t stream) {
  AS_REQUIRE(counts && u && rank_pos && rank_neg && is_pos && flag, AS_E_BADARG, "as_draw_distinct: null pointer");
  AS_REQUIRE(count_stride_g > 0 && count_stride_k > 0, AS_E_BADARG, "as_draw_distinct: count strides %d, %d", count_stride_g,
             count_stride_k);
  AS_REQUIRE(G > 0 && M > 0 && K > 0 && K <= 32 && M >= K, AS_E_UNSUPPORTED, "as_draw_distinct: K=%d of M=%d draws (K <= 32)", K, M);
  hipLaunchKernelGGL(draw_distinct_kernel, dim3(as_ceil_div(G, 64)), dim3(64), 0, (hipStream_t)stream, counts, count_stride_g,
                     count_stride_k, u, rank_pos, rank_neg, is_pos, flag, G, M, K);
  AS_CHECK_LAUNCH("draw_distinct");
  return AS_OK;
}

// =====================================================================================================
// Greedy grouping of merge_maps (stdroi:278-294) for all objects on the device: keep [G,P] (0/1), link [G,P,P] (0/1,
// cos >= thr) -> groups [G,P] int32 bit sets over the ORIGINAL prototype ids in the order the reference's loop emits
// them (0 = unused slot), ngroups [G].  Row i of the upper-triangular kept sub-matrix starts a group with every kept
// j >= i it links to, unless an earlier group already cleared row i (`sim_triu[weight > 0] *= 0`); columns are not
// cleared, exactly as in the reference.  One thread per object (P <= 32: a row is one word).
// =====================================================================================================
namespace {
__global__ void merge_plan_kernel(const uint8_t* __restrict__ keep, const uint8_t* __restrict__ link,
                                  int32_t* __restrict__ groups, int32_t* __restrict__ ngroups, int G, int P) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  unsigned kept = 0u;
  for (int p = 0; p < P; ++p) kept |= (keep[g * P + p] != 0 ? 1u : 0u) << p;
  unsigned cleared = 0u;
  int n = 0;
  for (int i = 0; i < P; ++i) {
    groups[g * P + i] = 0;
    if (!((kept >> i) & 1u) || ((cleared >> i) & 1u)) continue;
    unsigned row = 0u;
    for (int j = i; j < P; ++j) row |= (link[((size_t)g * P + i) * P + j] != 0 ? 1u : 0u) << j;
    row &= kept;
    if (row) { groups[g * P + n] = (int32_t)row; ++n; cleared |= row; }
  }
  ngroups[g] = n;
}
}  // namespace

extern "C" int as_merge_plan(const uint8_t* keep, const uint8_t* link, int32_t* groups, int32_t* ngroups, int G, int P,
                             as_stream_t stream) {
  AS_REQUIRE(keep && link && groups && ngroups, AS_E_BADARG, "as_merge_plan: null pointer");
  AS_REQUIRE(G > 0 && P > 0 && P <= 32, AS_E_UNSUPPORTED, "as_merge_plan: P=%d prototypes per object (max 32)", P);
  hipLaunchKernelGGL(merge_plan_kernel, dim3(as_ceil_div(G, 64)), dim3(64), 0, (hipStream_t)stream, keep, link, groups,
                     ngroups, G, P);
  AS_CHECK_LAUNCH("merge_plan");
  return AS_OK;
}

// =====================================================================================================
// merge_maps of stdroi:278-294 for every object in ONE launch: the cosine links between the kept prototypes, the greedy
// grouping above, and the merged prototypes  matmul(weight, prot) / (weight.sum(-1) + 1e-8)  of the first `slots` groups
// (unused slots are zero rows).  One workgroup per object; the object's prototypes are normalised into LDS
// (x / max(|x|, 1e-8), the reference's form), every (i, j >= i) cosine is one thread's dot product, thread 0 runs the
// grouping on the bit rows, and the members of a group are summed in index order.  *flag is OR-ed with 1 when an object
// has more than `slots` groups (the caller then takes its synchronous path).
// =====================================================================================================
namespace {
constexpr int MG_NT = 1024, MG_PMAX = 32;   // 16 waves: every phase is a short latency chain, so more waves = less time
__global__ __launch_bounds__(MG_NT) void merge_parts_kernel(const float* __restrict__ prot, const uint8_t* __restrict__ keep,
                                                            float thr, float* __restrict__ merged, int32_t* __restrict__ ngroups,
                                                            int32_t* __restrict__ flag, int P, int C, int slots) {
  extern __shared__ float mg_u[];                            // [P][C + 1] normalised prototypes (row pitch C + 1: no conflicts)
  __shared__ float nrm_s[MG_PMAX];
  __shared__ unsigned link_s[MG_PMAX], group_s[MG_PMAX];
  __shared__ uint8_t keep_s[MG_PMAX];
  __shared__ int n_s;
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pg = prot + (size_t)g * P * C;
  const int pitch = C + 1;
  // the object's prototypes -> LDS, eight independent loads per thread in flight (a load -> use loop of 60 trips per thread
  // is 60 serialised L2 round trips: that alone was 60 of the kernel's first version's 70 us)
  if (tid < MG_PMAX) link_s[tid] = 0u;
  if (tid < MG_PMAX) keep_s[tid] = tid < P ? keep[g * P + tid] : 0;       // (one thread reading them one by one: 20 round trips)
  {
    constexpr int U = 8;
    const int n = P * C;
    for (int i0 = tid; i0 < n; i0 += MG_NT * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = pg[min(i0 + u * MG_NT, n - 1)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * MG_NT;
        if (i < n) { const int p = i / C; mg_u[p * pitch + (i - p * C)] = v[u]; }
      }
    }
  }
  __syncthreads();
  for (int p = wave; p < P; p += MG_NT / 64) {                // |prot_p|: one wave per row, fixed shuffle tree
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) { const float v = mg_u[p * pitch + c]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    if (lane == 0) nrm_s[p] = fmaxf(sqrtf(s), 1e-8f);
  }
  __syncthreads();
  for (int i = tid; i < P * C; i += MG_NT) {                  // u = x / max(|x|, 1e-8), in place
    const int p = i / C, c = i - p * C;
    mg_u[p * pitch + c] = mg_u[p * pitch + c] / nrm_s[p];
  }
  __syncthreads();
  // cos(u_i, u_j) >= thr for j >= i (the upper triangle is all the grouping reads): one WAVE per pair, lanes along the
  // channels (conflict-free LDS rows), fixed shuffle tree -- a thread per pair walked 2 x C LDS words alone (70 us per launch)
  for (int pr = wave; pr < P * P; pr += MG_NT / 64) {
    const int i = pr / P, j = pr - i * P;
    if (j < i) continue;                                      // wave-uniform
    const float* a = mg_u + i * pitch;
    const float* b = mg_u + j * pitch;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;         // four reads in flight per operand (one LDS round trip per
    int c = lane;                                             // 4 terms instead of per term)
    for (; c + 192 < C; c += 256) {
      const float a0 = a[c], a1 = a[c + 64], a2 = a[c + 128], a3 = a[c + 192];
      const float b0 = b[c], b1 = b[c + 64], b2 = b[c + 128], b3 = b[c + 192];
      d0 = fmaf(a0, b0, d0); d1 = fmaf(a1, b1, d1); d2 = fmaf(a2, b2, d2); d3 = fmaf(a3, b3, d3);
    }
    for (; c < C; c += 64) d0 = fmaf(a[c], b[c], d0);
    const float d = wave_sum((d0 + d1) + (d2 + d3));
    if (lane == 0 && d >= thr) atomicOr(&link_s[i], 1u << j);
  }
  __syncthreads();
  if (tid == 0) {                                             // the greedy grouping of merge_plan_kernel
    unsigned kept = 0u;
    for (int p = 0; p < P; ++p) kept |= (keep_s[p] != 0 ? 1u : 0u) << p;
    unsigned cleared = 0u;
    int n = 0;
    for (int i = 0; i < P; ++i) {
      group_s[i] = 0u;
      if (!((kept >> i) & 1u) || ((cleared >> i) & 1u)) continue;
      const unsigned row = link_s[i] & kept;
      if (row) { group_s[n] = row; ++n; cleared |= row; }
    }
    n_s = n;
    ngroups[g] = n < slots ? n : slots;
    if (n > slots && flag != nullptr) atomicOr(flag, 1);
  }
  __syncthreads();
  // thread = channel: the P values of a channel are loaded ONCE, unconditionally and together (a load inside the member test
  // is a branch with its own s_waitcnt: ~100 dependent L2 round trips per thread, 90 us per launch), then every slot sums
  // its members in index order (adding +0 for the others leaves every partial sum bit-identical)
  for (int c = tid; c < C; c += MG_NT) {
    float v[MG_PMAX];
#pragma unroll
    for (int p = 0; p < MG_PMAX; ++p) v[p] = pg[(size_t)min(p, P - 1) * C + c];
    for (int sl = 0; sl < slots; ++sl) {
      const unsigned row = sl < P ? group_s[sl] : 0u;
      float acc = 0.0f, cnt = 0.0f;
#pragma unroll
      for (int p = 0; p < MG_PMAX; ++p) {
        const bool in = p < P && ((row >> p) & 1u);
        acc += in ? v[p] : 0.0f;
        cnt += in ? 1.0f : 0.0f;
      }
      merged[((size_t)g * slots + sl) * C + c] = acc / (cnt + 1e-8f);
    }
  }
}
}  // namespace

extern "C" int as_merge_parts(const float* prot, const uint8_t* keep, float thr, float* merged, int32_t* ngroups, int32_t* flag,
                              int G, int P, int C, int slots, as_stream_t stream) {
  AS_REQUIRE(prot && keep && merged && ngroups, AS_E_BADARG, "as_merge_parts: null pointer");
  AS_REQUIRE(G > 0 && P > 0 && P <= MG_PMAX && C > 0 && slots > 0 && slots <= P, AS_E_UNSUPPORTED,
             "as_merge_parts: P=%d prototypes per object (max %d), %d slots", P, MG_PMAX, slots);
  const size_t lds = (size_t)P * (C + 1) * sizeof(float);
  AS_REQUIRE(lds <= 140 * 1024, AS_E_UNSUPPORTED, "as_merge_parts: %d x %d prototypes exceed the LDS image", P, C);
  // (a driver call in the latency-critical per-image chain: made only when this launch needs more than any before it; a race
  //  between two host threads only repeats the idempotent call -- ADVICE r05)
  static std::atomic<size_t> lds_set{48 * 1024};
  if (lds > lds_set.load(std::memory_order_relaxed)) {
    (void)hipFuncSetAttribute((const void*)merge_parts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    size_t seen = lds_set.load(std::memory_order_relaxed);
    while (seen < lds && !lds_set.compare_exchange_weak(seen, lds, std::memory_order_relaxed)) {}
  }
  hipLaunchKernelGGL(merge_parts_kernel, dim3(G), dim3(MG_NT), lds, (hipStream_t)stream, prot, keep, thr, merged, ngroups, flag,
                     P, C, slots);
  AS_CHECK_LAUNCH("merge_parts");
  return AS_OK;
}

// =====================================================================================================
// The layer choice of the stand-in selector and everything indexed by it, one thread per object (the MIL head's place in
// stdroi:2953-2972 taken by the depth whose CAM box has the median area, roi_head.median_area_selector): stable ascending
// rank of the Lc box areas, the chosen box, its row in the layer-major CAM stack and its patch box (`rois // stride`,
// stdroi:1812).  Replaces ~20 tensor ops of index arithmetic per step.
// =====================================================================================================
namespace {
__global__ __launch_bounds__(64) void select_median_boxes_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ meta, int Lc, float stride,
                                           const int64_t* __restrict__ pick_in, int64_t* __restrict__ pick,
                                           float* __restrict__ chosen,
                                           int32_t* __restrict__ map_idx, int32_t* __restrict__ box_patch,
                                           int32_t* __restrict__ box_int, const int32_t* __restrict__ status,
                                           int32_t* __restrict__ bad, int n) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  const int off = meta[3 * o], cnt = meta[3 * o + 1], g = meta[3 * o + 2];
  const float4* b = reinterpret_cast<const float4*>(boxes);
  const int want = (Lc - 1) / 2;
  int sel = pick_in ? min(max((int)pick_in[o], 0), Lc - 1) : 0;      // (a selector's own choice: only the indexing is done here)
  if (!pick_in && Lc <= 16) {
    // all box loads in flight at once, then the Lc^2 compares on this thread's column of an LDS table (a load inside the
    // rank loops was one L2 round trip per compare: 11 us for 7 layers; 256 unrolled compares on registers: 4000
    // instructions for one wave)
    __shared__ float sa[16][64];
    const int t = threadIdx.x;
    float4 bb[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) bb[l] = b[off + min(l, Lc - 1) * cnt + g];
#pragma unroll
    for (int l = 0; l < 16; ++l) sa[l][t] = fmaxf(bb[l].z - bb[l].x, 0.f) * fmaxf(bb[l].w - bb[l].y, 0.f);
    for (int l = 0; l < Lc; ++l) {
      const float al = sa[l][t];
      int rank = 0;
      for (int m = 0; m < Lc; ++m) {
        const float am = sa[m][t];
        rank += (am < al) || (am == al && m < l);
      }
      if (rank == want) sel = l;
    }
  } else {
    for (int l = 0; l < Lc && !pick_in; ++l) {
      const float4 bl = b[off + l * cnt + g];
      const float al = fmaxf(bl.z - bl.x, 0.f) * fmaxf(bl.w - bl.y, 0.f);
      int rank = 0;
      for (int m = 0; m < Lc; ++m) {
        const float4 bm = b[off + m * cnt + g];
        const float am = fmaxf(bm.z - bm.x, 0.f) * fmaxf(bm.w - bm.y, 0.f);
        rank += (am < al) || (am == al && m < l);
      }
      if (rank == want) sel = l;
    }
  }
  const int row = off + sel * cnt + g;
  const float4 c = b[row];
  pick[o] = sel;
  reinterpret_cast<float4*>(chosen)[o] = c;
  map_idx[o] = row;
  reinterpret_cast<int4*>(box_patch)[o] = make_int4((int)floorf(c.x / stride), (int)floorf(c.y / stride),
                                                    (int)floorf(c.z / stride), (int)floorf(c.w / stride));
  if (box_int) reinterpret_cast<int4*>(box_int)[o] = make_int4((int)c.x, (int)c.y, (int)c.z, (int)c.w);
  if (status) {                                              // every row of `boxes` belongs to exactly one object
    int worst = 1;                                            // (no short-circuit: the loads are independent)
    for (int l = 0; l < Lc; ++l) worst = min(worst, status[off + l * cnt + g]);
    if (worst <= 0) atomicOr(bad, 1);
  }
}
}  // namespace

extern "C" int as_select_median_boxes(const float* boxes, const int32_t* meta, int Lc, int stride, const int64_t* pick_in,
                                      int64_t* pick, float* chosen,
                                      int32_t* map_idx, int32_t* box_patch, int32_t* box_int, const int32_t* status,
                                      int32_t* bad, int n, as_stream_t stream) {
  AS_REQUIRE(boxes && meta && pick && chosen && map_idx && box_patch, AS_E_BADARG, "as_select_median_boxes: null pointer");
  AS_REQUIRE(n > 0 && Lc > 0 && Lc <= 64 && stride > 0, AS_E_UNSUPPORTED, "as_select_median_boxes: n=%d Lc=%d stride=%d", n, Lc, stride);
  AS_REQUIRE(!status || bad, AS_E_BADARG, "as_select_median_boxes: status without a flag to raise");
  hipLaunchKernelGGL(select_median_boxes_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, boxes, meta, Lc,
                     (float)stride, pick_in, pick, chosen, map_idx, box_patch, box_int, status, bad, n);
  AS_CHECK_LAUNCH("select_median_boxes");
  return AS_OK;
}

// =====================================================================================================
// Patch-grid foreground of get_semantic_centers (stdroi:2011-2012, 2020): erode(map_fg > thr, k) at full resolution,
// bilinear DOWN to the patch grid.  For an exact integer factor `up` the down-sampled value is the mean of the 2x2
// pixels (up/2-1, up/2) of the patch with weights 0.5/0.5, so only those four erosions are evaluated: a 16-lane group
// per patch, lane = one row of the (k+1) x (k+1) block the four windows span, rows combined with AND shuffles.
// Also the binary patch map (fg_inter > thr) and its per-object count (the grid-seed candidates, :1784).
// =====================================================================================================
namespace {

__global__ __launch_bounds__(RF_NT) void semantic_prestage_kernel(const float* __restrict__ map_fg, float thr, int G,
                                                                  int Hp, int Wp, int up, int k,
                                                                  float* __restrict__ fg_inter, uint8_t* __restrict__ mask,
                                                                  int32_t* __restrict__ counts) {
  const int H = Hp * up, W = Wp * up, Np = Hp * Wp, r0 = k / 2;
  const int gid = blockIdx.x * RF_NT + threadIdx.x;
  const int patch = gid >> 4, r = gid & 15;                  // r = row of the block (k + 1 <= 16 rows)
  const bool live = patch < G * Np;
  const int pc = live ? patch : 0;
  const int g = pc / Np, pp = pc - g * Np, py = pp / Wp, px = pp - py * Wp;
  const int y = up * py + up / 2 - 1 - r0 + r, xb = up * px + up / 2 - 1 - r0;
  const float* row = map_fg + ((size_t)g * H + min(max(y, 0), H - 1)) * W;
  const bool yin = y >= 0 && y < H && r <= k;
  unsigned bits = 0;                                          // bit c: pixel (y, xb + c) passes or lies outside the image
  for (int c = 0; c <= k; ++c) {
    const int x = xb + c;
    const bool in = yin && x >= 0 && x < W;
    const float v = row[min(max(x, 0), W - 1)];
    if (!in || v > thr) bits |= 1u << c;
  }
  const unsigned full = (1u << k) - 1u;
  float val = 0.0f;
#pragma unroll
  for (int sr = 0; sr < 2; ++sr)
#pragma unroll
    for (int sc = 0; sc < 2; ++sc) {
      // window of sample (sr, sc): rows sr .. sr+k-1, columns sc .. sc+k-1 of the block
      unsigned ok = (r < sr || r > sr + k - 1) ? 1u : (((bits >> sc) & full) == full ? 1u : 0u);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) ok &= __shfl_xor(ok, o);
      const float e = ok ? 1.0f : 0.0f;
      // 0.5 * (0.5 a + 0.5 b) + 0.5 * (0.5 c + 0.5 d): exact for 0/1 inputs
      val += 0.25f * e;
    }
  const bool m = val > thr;
  if (live && r == 0) {
    fg_inter[pc] = val;
    mask[pc] = m ? 1 : 0;
  }
  // objects do not straddle waves unless Np % 4 != 0: count per lane group leader with one atomic per wave and object
  const unsigned long long b = __ballot(live && r == 0 && m);
  if (b) {
    const int lane = threadIdx.x & 63;
    const int g_first = __shfl(g, 0), g_last = __shfl(g, 48);
    if (g_first == g_last) {
      if (lane == 0 && counts) atomicAdd(&counts[g_first], __popcll(b));
    } else if (live && r == 0 && m && counts) {
      atomicAdd(&counts[g], 1);
    }
  }
}

}  // namespace

extern "C" int as_semantic_prestage(const float* map_fg, float thr, int k, int G, int Hp, int Wp, int up, float* fg_inter,
                                    uint8_t* mask, int32_t* counts, as_stream_t stream) {
  AS_REQUIRE(map_fg && fg_inter && mask, AS_E_BADARG, "as_semantic_prestage: null pointer");
  AS_REQUIRE(G > 0 && Hp > 0 && Wp > 0 && up >= 2 && up % 2 == 0 && k >= 1 && (k & 1) == 1 && k <= 15, AS_E_UNSUPPORTED,
             "as_semantic_prestage: even scale >= 2 and odd erosion size <= 15 (got %d, %d)", up, k);
  hipStream_t s = (hipStream_t)stream;
  if (counts) (void)hipMemsetAsync(counts, 0, (size_t)G * 4, s);     // (NULL: a caller that does not read the counts)
  const size_t threads = (size_t)G * Hp * Wp * 16;
  hipLaunchKernelGGL(semantic_prestage_kernel, dim3((unsigned)((threads + RF_NT - 1) / RF_NT)), dim3(RF_NT), 0, s, map_fg,
                     thr, G, Hp, Wp, up, k, fg_inter, mask, counts);
  AS_CHECK_LAUNCH("semantic_prestage");
  return AS_OK;
}

// =====================================================================================================
// Rank select: the k-th set pixel (raster order, = the order of torch's .nonzero()) of byte masks, used to turn the
// reference's `coords[random_index]` (stdroi:368-369, :456) into a lookup that needs no compaction pass.
//   counts   grid (chunks, M): set bytes per 4096-byte chunk
//   select   grid (K, M), one wave per rank: prefix over the chunk counts, then inside the chunk
// =====================================================================================================
namespace {

constexpr int RS_CHUNK = 4096;

__device__ __forceinline__ int bytesum16(const uint8_t* p) {          // p 16-byte aligned, bytes are 0/1
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  return (int)(((v.x * 0x01010101u) >> 24) + ((v.y * 0x01010101u) >> 24) + ((v.z * 0x01010101u) >> 24) +
               ((v.w * 0x01010101u) >> 24));
}

__global__ __launch_bounds__(RF_NT) void rank_counts_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ cc,
                                                            int HW, int nchunk) {
  __shared__ int sh[RF_NT];
  const int c = blockIdx.x, m = blockIdx.y, tid = threadIdx.x;
  const size_t base = (size_t)m * HW + (size_t)c * RS_CHUNK;
  const int off = tid * 16;
  int v = 0;
  if (c * RS_CHUNK + off + 16 <= HW) {
    v = bytesum16(mask + base + off);
  } else {
    for (int i = 0; i < 16; ++i)
      if (c * RS_CHUNK + off + i < HW) v += mask[base + off + i] != 0;
  }
  sh[tid] = v;
  __syncthreads();
  for (int o = RF_NT / 2; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  if (tid == 0) cc[(size_t)m * nchunk + c] = sh[0];
}

// number of set bytes of each row of a [M, HW] 0/1 byte mask (torch's bool -> int64 row sum takes 70-220 us here)
__global__ __launch_bounds__(RF_NT) void mask_count_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ counts,
                                                           int HW) {
  __shared__ int sh[RF_NT];
  const int m = blockIdx.y, tid = threadIdx.x;
  const uint8_t* row = mask + (size_t)m * HW;
  int v = 0;
  const int n16 = HW / 16;
  for (int i = blockIdx.x * RF_NT + tid; i < n16; i += gridDim.x * RF_NT) v += bytesum16(row + (size_t)i * 16);
  if (blockIdx.x == 0)
    for (int i = n16 * 16 + tid; i < HW; i += RF_NT) v += row[i] != 0;
  sh[tid] = v;
  __syncthreads();
  for (int o = RF_NT / 2; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  if (tid == 0 && sh[0] != 0) atomicAdd(&counts[m], sh[0]);
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane) {
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  return inc - v;
}

// result of one (row, rank): the flat index, and / or its (x, y) [or (y, x)] coordinates on a W-wide grid as int64 with an
// out-of-range rank (-1) clamped to pixel 0 -- what the callers' `clamp(min=0)`, `% W`, `// W`, `stack` chain produced
// (as_rank_draw_xy only) patch: the index of the selected pixel's patch on a patch_w-wide grid of patch_div-pixel patches,
// plus patch_base, written to row (m + M - row_rot) % M of patch_out -- what the callers' `// 16`, `clamp`, `cat` chain and
// the gather index of the seed features were built from
struct RankPatch { long long* out; int div, w, rot; long long base; };
__device__ __forceinline__ void rank_select_store(int32_t* out, long long* out_xy, size_t idx, int flat, int W, int yx,
                                                  const RankPatch* rp = nullptr, int m = 0, int M = 1, int k = 0, int K = 1) {
  if (out != nullptr) out[idx] = flat;
  const int f = flat < 0 ? 0 : flat, y = f / W, x = f - y * W;
  if (out_xy != nullptr) {
    out_xy[2 * idx + 0] = yx ? y : x;
    out_xy[2 * idx + 1] = yx ? x : y;
  }
  if (rp != nullptr && rp->out != nullptr)
    rp->out[(size_t)((m + M - rp->rot) % M) * K + k] = rp->base + (long long)(y / rp->div) * rp->w + x / rp->div;
}

// MODE 0: the rank comes from `ranks`.  MODE 1 / 2 (as_rank_draw_xy): the rank is derived from the row's population n, which
// the chunk counts already give -- 1: a uniform draw, min(int(u[m][k] * float(n)), max(n - 1, 0)), the fast-RNG form of the
// reference's `torch.randint(n, ...)` (stdroi:366); 2: the k-th of K grid-strided positives, k * max(n / K, 1) (stdroi:1790-
// 1792).  Both raise *flag when n < K (the reference's refill branches, which the caller redoes on the host path).
template <int MODE>
__global__ __launch_bounds__(64) void rank_select_kernel(const uint8_t* __restrict__ mask, const int32_t* __restrict__ cc,
                                                         const int32_t* __restrict__ ranks, const float* __restrict__ u,
                                                         int32_t* __restrict__ flag, int32_t* __restrict__ out,
                                                         long long* __restrict__ out_xy, int W, int yx, int HW, int nchunk,
                                                         int K, RankPatch rpv) {
  const int k = blockIdx.x, m = blockIdx.y, lane = threadIdx.x;
  const RankPatch* rp = MODE == 0 ? nullptr : &rpv;
  const int M = gridDim.y;
  int r = MODE == 0 ? ranks[(size_t)m * K + k] : 0;
  const float uk = MODE == 1 ? u[(size_t)m * K + k] : 0.0f;
  const int32_t* c = cc + (size_t)m * nchunk;
  // level 1: which chunk
  const int cpl = (nchunk + 63) / 64;
  int mine = 0;
  int c4[4] = {0, 0, 0, 0};
  const bool vec = cpl == 4 && nchunk % 4 == 0 && (lane + 1) * 4 <= nchunk;   // 1024^2 masks: 256 chunks, an int4 per lane
  if (vec) {
    const int4 v = *reinterpret_cast<const int4*>(c + lane * 4);
    c4[0] = v.x; c4[1] = v.y; c4[2] = v.z; c4[3] = v.w;
    mine = (v.x + v.y) + (v.z + v.w);
  } else {
    for (int i = lane * cpl; i < min((lane + 1) * cpl, nchunk); ++i) mine += c[i];
  }
  const int before = wave_excl_scan(mine, lane);
  if (MODE != 0) {
    const int n = __shfl(before + mine, 63);       // the row's population
    if (MODE == 1) r = min((int)(uk * (float)n), max(n - 1, 0));
    else r = k * max(n / K, 1);
    if (k == 0 && lane == 0 && n < K && flag != nullptr) atomicOr(flag, 1);
  }
  const unsigned long long hit = __ballot(r >= before && r < before + mine);
  if (hit == 0ull || r < 0) {                    // rank beyond the population
    if (lane == 0) rank_select_store(out, out_xy, (size_t)m * K + k, -1, W, yx, rp, m, M, k, K);
    return;
  }
  const int owner = __ffsll((long long)hit) - 1;
  int chunk = -1, rem = 0;
  if (lane == owner) {
    int acc = before;
    if (vec) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (chunk < 0 && r < acc + c4[j]) { chunk = lane * 4 + j; rem = r - acc; }
        acc += c4[j];
      }
    } else {
      for (int i = lane * cpl; i < min((lane + 1) * cpl, nchunk); ++i) {
        if (r < acc + c[i]) { chunk = i; rem = r - acc; break; }
        acc += c[i];
      }
    }
  }
  chunk = __shfl(chunk, owner);
  rem = __shfl(rem, owner);
  // level 2: inside the chunk, 64 bytes per lane held in registers as a 64-bit occupancy word
  const size_t base = (size_t)m * HW + (size_t)chunk * RS_CHUNK;
  const int off = lane * 64;
  unsigned long long occ = 0ull;
  if (chunk * RS_CHUNK + off + 64 <= HW) {
    const uint4* p4 = reinterpret_cast<const uint4*>(mask + base + off);      // rows and chunks are 16-byte aligned
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 w = p4[j];
      const unsigned xs[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // 4 bytes of 0/1 -> 4 bits (multiplying gathers byte i's bit 0 at bit 24 + i; the partial products do not carry)
        const unsigned long long nib = ((xs[q] & 0x01010101u) * 0x01020408u) >> 24 & 0xfu;
        occ |= nib << (16 * j + 4 * q);
      }
    }
  } else {
    for (int i = 0; i < 64; ++i)
      if (chunk * RS_CHUNK + off + i < HW && mask[base + off + i] != 0) occ |= 1ull << i;
  }
  const int cnt = __popcll(occ);
  const int b2 = wave_excl_scan(cnt, lane);
  const unsigned long long hit2 = __ballot(rem >= b2 && rem < b2 + cnt);
  const int owner2 = __ffsll((long long)hit2) - 1;
  if (lane == owner2) {
    int left = rem - b2, pos = 0;                 // position of the left-th set bit of occ
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      const int c = __popcll(occ & ((1ull << sft) - 1ull));
      if (left >= c) { occ >>= sft; left -= c; pos += sft; }
    }
    rank_select_store(out, out_xy, (size_t)m * K + k, chunk * RS_CHUNK + off + pos, W, yx, rp, m, M, k, K);
  }
}

}  // namespace

// =====================================================================================================
// All three candidate masks of one image's objects in one call (stdroi:433-461 mask points, :2356-2358 pseudo masks):
//   pos    = erode_k(in_crop && map_fg > max_crop(map_fg) * pos_thr)   (foreground point candidates, :442)
//   neg    =          in_crop && map_bg > max_crop(map_bg) * neg_thr   (background point candidates, :443)
//   pseudo =                     map_fg > max(map_fg)      * mask_thr   (the pseudo mask, :2357)
// The three maxima come from one pass over the maps, the three thresholdings from a second; the separable erosion
// reuses the crop kernels above.  counts [3, G]: set pixels of pos / neg / pseudo.
// =====================================================================================================
namespace {

template <bool VEC>
__global__ __launch_bounds__(RF_NT) void cand_max_kernel(const float* __restrict__ map_fg, const float* __restrict__ map_bg,
                                                         const int32_t* __restrict__ crops, unsigned* __restrict__ mx,
                                                         int32_t* __restrict__ counts_zero, int G, int H, int W) {
  __shared__ float sh[RF_NT];
  const int g = blockIdx.y;
  // the candidate counts are accumulated by the kernels launched AFTER this one: cleared here instead of by a fill launch
  if (blockIdx.x == 0 && threadIdx.x < 3) counts_zero[threadIdx.x * G + g] = 0;
  const Crop c = load_crop(crops, g, H, W);
  const size_t base = (size_t)g * H * W;
  float vf = -INFINITY, vb = -INFINITY, va = -INFINITY;
  if (VEC) {                                             // W % 4 == 0: four pixels of one row per 16-byte load
    const int n4 = H * W / 4;
    for (int i4 = blockIdx.x * RF_NT + threadIdx.x; i4 < n4; i4 += gridDim.x * RF_NT) {
      const int i = i4 * 4, y = i / W, x = i - y * W;
      const float4 f = *reinterpret_cast<const float4*>(map_fg + base + i);
      va = fmaxf(va, fmaxf(fmaxf(f.x, f.y), fmaxf(f.z, f.w)));
      if (y >= c.y0 && y < c.y1 && x + 3 >= c.x0 && x < c.x1) {
        const float4 b = *reinterpret_cast<const float4*>(map_bg + base + i);
        const float fe[4] = {f.x, f.y, f.z, f.w}, be[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (x + j >= c.x0 && x + j < c.x1) { vf = fmaxf(vf, fe[j]); vb = fmaxf(vb, be[j]); }
      }
    }
  } else {
    for (int i = blockIdx.x * RF_NT + threadIdx.x; i < H * W; i += gridDim.x * RF_NT) {
      const int y = i / W, x = i - y * W;
      const float f = map_fg[base + i];
      va = fmaxf(va, f);
      if (x >= c.x0 && x < c.x1 && y >= c.y0 && y < c.y1) {
        vf = fmaxf(vf, f);
        vb = fmaxf(vb, map_bg[base + i]);
      }
    }
  }
  const float rf = block_max(vf, sh), rb = block_max(vb, sh), ra = block_max(va, sh);
  if (threadIdx.x == 0) {
    if (rf > -INFINITY) atomicMax(&mx[g], f2ord(rf));
    if (rb > -INFINITY) atomicMax(&mx[G + g], f2ord(rb));
    atomicMax(&mx[2 * G + g], f2ord(ra));
  }
}

template <bool VEC>
__global__ __launch_bounds__(RF_NT) void cand_threshold_kernel(const float* __restrict__ map_fg,
                                                               const float* __restrict__ map_bg,
                                                               const int32_t* __restrict__ crops,
                                                               const unsigned* __restrict__ mx, float pos_thr,
                                                               float neg_thr, float mask_thr, uint8_t* __restrict__ t0,
                                                               uint8_t* __restrict__ neg, uint8_t* __restrict__ pseudo,
                                                               int32_t* __restrict__ counts, int G, int H, int W) {
  __shared__ int shc[RF_NT];
  const int g = blockIdx.y;
  const Crop c = load_crop(crops, g, H, W);
  const size_t base = (size_t)g * H * W;
  const float tp = ord2f(mx[g]) * pos_thr, tn = ord2f(mx[G + g]) * neg_thr, ta = ord2f(mx[2 * G + g]) * mask_thr;
  int cn = 0, ca = 0;
  if (VEC) {
    const int n4 = H * W / 4;
    for (int i4 = blockIdx.x * RF_NT + threadIdx.x; i4 < n4; i4 += gridDim.x * RF_NT) {
      const int i = i4 * 4, y = i / W, x = i - y * W;
      const float4 f = *reinterpret_cast<const float4*>(map_fg + base + i);
      const bool rowin = y >= c.y0 && y < c.y1 && x + 3 >= c.x0 && x < c.x1;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rowin) b = *reinterpret_cast<const float4*>(map_bg + base + i);
      const float fe[4] = {f.x, f.y, f.z, f.w}, be[4] = {b.x, b.y, b.z, b.w};
      unsigned vp4 = 0, vn4 = 0, va4 = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool inside = rowin && x + j >= c.x0 && x + j < c.x1;
        const bool vp = inside && fe[j] > tp, vn = inside && be[j] > tn, va = fe[j] > ta;
        vp4 |= (vp ? 1u : 0u) << (8 * j); vn4 |= (vn ? 1u : 0u) << (8 * j); va4 |= (va ? 1u : 0u) << (8 * j);
        cn += vn ? 1 : 0; ca += va ? 1 : 0;
      }
      *reinterpret_cast<unsigned*>(t0 + base + i) = vp4;
      *reinterpret_cast<unsigned*>(neg + base + i) = vn4;
      *reinterpret_cast<unsigned*>(pseudo + base + i) = va4;
    }
  } else {
    for (int i = blockIdx.x * RF_NT + threadIdx.x; i < H * W; i += gridDim.x * RF_NT) {
      const int y = i / W, x = i - y * W;
      const bool inside = x >= c.x0 && x < c.x1 && y >= c.y0 && y < c.y1;
      const float f = map_fg[base + i];
      const bool vp = inside && f > tp;
      const bool vn = inside && map_bg[base + i] > tn;
      const bool va = f > ta;
      t0[base + i] = vp ? 1 : 0;
      neg[base + i] = vn ? 1 : 0;
      pseudo[base + i] = va ? 1 : 0;
      cn += vn ? 1 : 0; ca += va ? 1 : 0;
    }
  }
  for (int which = 0; which < 2; ++which) {
    shc[threadIdx.x] = which == 0 ? cn : ca;
    __syncthreads();
    for (int o = RF_NT / 2; o > 0; o >>= 1) {
      if (threadIdx.x < o) shc[threadIdx.x] += shc[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0 && shc[0] > 0) atomicAdd(&counts[(1 + which) * G + g], shc[0]);
    __syncthreads();
  }
}

}  // namespace

extern "C" size_t as_mask_candidates_workspace_bytes(int G, int H, int W) {
  if (G <= 0 || H <= 0 || W <= 0) return 0;
  return 2 * (((size_t)G * H * W + 255) / 256 * 256) + ((size_t)3 * G * 4 + 255) / 256 * 256;
}

extern "C" int as_mask_candidates(const float* map_fg, const float* map_bg, const int32_t* crops, float pos_thr,
                                  float neg_thr, float mask_thr, int k, uint8_t* pos, uint8_t* neg, uint8_t* pseudo,
                                  int32_t* counts, void* ws, size_t ws_bytes, int G, int H, int W, as_stream_t stream) {
  AS_REQUIRE(map_fg && map_bg && crops && pos && neg && pseudo && counts && ws, AS_E_BADARG,
             "as_mask_candidates: null pointer");
  AS_REQUIRE(G > 0 && H > 0 && W > 0 && k >= 1 && (k & 1) == 1, AS_E_BADARG, "as_mask_candidates: bad sizes (k odd)");
  AS_REQUIRE(ws_bytes >= as_mask_candidates_workspace_bytes(G, H, W), AS_E_WORKSPACE,
             "as_mask_candidates: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t plane = ((size_t)G * H * W + 255) / 256 * 256;
  uint8_t* t0 = (uint8_t*)ws;
  uint8_t* t1 = t0 + plane;
  unsigned* mx = (unsigned*)(t1 + plane);
  const int bx = (int)(((size_t)H * W + RF_NT * 4 - 1) / (RF_NT * 4));
  (void)hipMemsetAsync(mx, 0, (size_t)3 * G * 4, s);            // (counts [3,G]: cleared by cand_max_kernel)
  // atomics per workgroup on a few words per object: keep the workgroup count per object small
  // 16-byte accesses when rows are a multiple of 4 pixels and every plane (maps and byte masks, G*H*W apart) stays aligned
  const bool vec = W % 4 == 0 && (((size_t)map_fg | (size_t)map_bg) % 16 == 0) &&
                   (((size_t)(k == 1 ? pos : t0) | (size_t)neg | (size_t)pseudo) % 4 == 0);
  if (vec) {
    hipLaunchKernelGGL(cand_max_kernel<true>, dim3(bx < 256 ? bx : 256, G), dim3(RF_NT), 0, s, map_fg, map_bg, crops, mx, counts, G, H, W);
    hipLaunchKernelGGL(cand_threshold_kernel<true>, dim3(bx < 256 ? bx : 256, G), dim3(RF_NT), 0, s, map_fg, map_bg, crops, mx,
                       pos_thr, neg_thr, mask_thr, k == 1 ? pos : t0, neg, pseudo, counts, G, H, W);
  } else {
    hipLaunchKernelGGL(cand_max_kernel<false>, dim3(bx < 256 ? bx : 256, G), dim3(RF_NT), 0, s, map_fg, map_bg, crops, mx, counts, G, H, W);
    hipLaunchKernelGGL(cand_threshold_kernel<false>, dim3(bx < 96 ? bx : 96, G), dim3(RF_NT), 0, s, map_fg, map_bg, crops, mx,
                       pos_thr, neg_thr, mask_thr, k == 1 ? pos : t0, neg, pseudo, counts, G, H, W);
  }
  if (k == 1) {
    // pos needs its count: a plain count of the thresholded crop
    hipLaunchKernelGGL(mask_count_kernel, dim3(64, G), dim3(RF_NT), 0, s, pos, counts, H * W);
  } else {
    const int r = k / 2;
    hipLaunchKernelGGL((crop_mask_kernel<1, false>), dim3(bx, G), dim3(RF_NT), 0, s, map_fg, t0, crops, mx, 0.0f, 0, r, t1,
                       counts, H, W);
    hipLaunchKernelGGL((crop_mask_kernel<2, true>), dim3(bx, G), dim3(RF_NT), 0, s, map_fg, t1, crops, mx, 0.0f, 0, r, pos,
                       counts, H, W);
  }
  AS_CHECK_LAUNCH("mask_candidates");
  return AS_OK;
}

extern "C" size_t as_rank_select_workspace_bytes(int M, int HW) {
  if (M <= 0 || HW <= 0) return 0;
  return ((size_t)M * as_ceil_div(HW, RS_CHUNK) * 4 + 255) / 256 * 256;
}

static int rank_select_launch(const uint8_t* mask, const int32_t* ranks, int32_t* out, long long* out_xy, int W, int yx, void* ws,
                              size_t ws_bytes, int M, int HW, int K, as_stream_t stream, const char* who, int mode = 0,
                              const float* u = nullptr, int32_t* flag = nullptr, RankPatch rp = RankPatch{nullptr, 1, 1, 0, 0}) {
  AS_REQUIRE(mask && (ranks || mode != 0) && (out || out_xy || rp.out) && ws, AS_E_BADARG, "%s: null pointer", who);
  AS_REQUIRE(M > 0 && HW > 0 && K > 0 && HW % 16 == 0 && W > 0, AS_E_BADARG, "%s: bad sizes (HW %% 16 == 0)", who);
  AS_REQUIRE(ws_bytes >= as_rank_select_workspace_bytes(M, HW), AS_E_WORKSPACE, "%s: workspace too small", who);
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = as_ceil_div(HW, RS_CHUNK);
  int32_t* cc = (int32_t*)ws;
  hipLaunchKernelGGL(rank_counts_kernel, dim3(nchunk, M), dim3(RF_NT), 0, s, mask, cc, HW, nchunk);
  if (mode == 1)
    hipLaunchKernelGGL(rank_select_kernel<1>, dim3(K, M), dim3(64), 0, s, mask, cc, ranks, u, flag, out, out_xy, W, yx, HW, nchunk, K, rp);
  else if (mode == 2)
    hipLaunchKernelGGL(rank_select_kernel<2>, dim3(K, M), dim3(64), 0, s, mask, cc, ranks, u, flag, out, out_xy, W, yx, HW, nchunk, K, rp);
  else
    hipLaunchKernelGGL(rank_select_kernel<0>, dim3(K, M), dim3(64), 0, s, mask, cc, ranks, u, flag, out, out_xy, W, yx, HW, nchunk, K, rp);
  AS_CHECK_LAUNCH(who);
  return AS_OK;
}

extern "C" int as_rank_select(const uint8_t* mask, const int32_t* ranks, int32_t* out, void* ws, size_t ws_bytes, int M,
                              int HW, int K, as_stream_t stream) {
  return rank_select_launch(mask, ranks, out, nullptr, 1, 0, ws, ws_bytes, M, HW, K, stream, "as_rank_select");
}

extern "C" int as_rank_select_xy(const uint8_t* mask, const int32_t* ranks, int64_t* out_xy, void* ws, size_t ws_bytes, int M,
                                 int HW, int K, int W, int yx_order, as_stream_t stream) {
  return rank_select_launch(mask, ranks, nullptr, (long long*)out_xy, W, yx_order ? 1 : 0, ws, ws_bytes, M, HW, K, stream,
                            "as_rank_select_xy");
}

extern "C" int as_rank_draw_xy(const uint8_t* mask, int mode, const float* u, int32_t* flag, int64_t* out_xy, int64_t* patch_out,
                               int patch_div, int patch_w, int64_t patch_base, int row_rot, void* ws, size_t ws_bytes, int M,
                               int HW, int K, int W, int yx_order, as_stream_t stream) {
  AS_REQUIRE(mode == 1 || mode == 2, AS_E_BADARG, "as_rank_draw_xy: mode %d (1 = uniform draws, 2 = grid stride)", mode);
  AS_REQUIRE(mode != 1 || u, AS_E_BADARG, "as_rank_draw_xy: mode 1 needs the uniform numbers");
  AS_REQUIRE(out_xy || patch_out, AS_E_BADARG, "as_rank_draw_xy: null pointer");
  AS_REQUIRE(!patch_out || (patch_div > 0 && patch_w > 0 && row_rot >= 0 && row_rot < M), AS_E_BADARG,
             "as_rank_draw_xy: patch grid (div %d, width %d, row rotation %d of %d rows)", patch_div, patch_w, row_rot, M);
  const RankPatch rp{(long long*)patch_out, patch_div > 0 ? patch_div : 1, patch_w > 0 ? patch_w : 1, row_rot, (long long)patch_base};
  return rank_select_launch(mask, nullptr, nullptr, (long long*)out_xy, W, yx_order ? 1 : 0, ws, ws_bytes, M, HW, K, stream,
                            "as_rank_draw_xy", mode, u, flag, rp);
}

extern "C" int as_mask_count(const uint8_t* mask, int32_t* counts, int M, int HW, as_stream_t stream) {
  AS_REQUIRE(mask && counts, AS_E_BADARG, "as_mask_count: null pointer");
  AS_REQUIRE(M > 0 && HW > 0 && ((size_t)mask % 16) == 0 && (HW % 16 == 0 || M == 1), AS_E_BADARG,
             "as_mask_count: rows must be 16-byte aligned (HW %% 16 == 0)");
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(counts, 0, (size_t)M * sizeof(int32_t), s);
  const int bx = as_ceil_div(HW / 16 + 1, RF_NT);
  hipLaunchKernelGGL(mask_count_kernel, dim3(bx < 32 ? bx : 32, M), dim3(RF_NT), 0, s, mask, counts, HW);
  AS_CHECK_LAUNCH("mask_count");
  return AS_OK;
}
