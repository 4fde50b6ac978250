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
