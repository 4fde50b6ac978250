// RoIAlign forward / backward on the stride-16 feature map: the 7x7 / 14x14 RoI features the MIL, box and
// mask heads consume (reference: mmcv.ops.RoIAlign as configured at configs/mae/attnshift_voc12aug.py:64-68, 123-127 --
// adaptive sampling_ratio=0, aligned=True, average pooling; called from stdroi:2958 `bbox_roi_extractor` and the
// standard bbox / mask forward of the RoI head).  mmcv-full 1.3.8 is not part of the reference tree: the kernel
// follows its published algorithm (parity unpinned) and is tested against the tensor-op restatement in
// attentionshift_amd/mil_head.py.
//
// Layout is chosen for HBM, not inherited: features are read TOKEN-MAJOR [B, H, W, C] (the layout the ViT produces
// them in) and RoI features are written [R, out*out, C] -- exactly the token sequence the MAE-decoder heads flatten to --
// so every load and store of a workgroup is a contiguous run of channels (float4 per lane).  One workgroup per output
// bin (roi, ph, pw); the tensor-op form materialised [R, C, H, W] + [R, C, out*g, W] (several GB at 1024 RoIs x 768
// channels).  The backward is a deterministic gather (below).
#include "common.h"

namespace {

struct RoiGeom { int b, gw, gh; float x1, y1, bw, bh; bool empty; };

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ rois, int r, float scale, int out, int sampling_ratio,
                                            int aligned, int B) {
  const float* p = rois + (size_t)r * 5;
  RoiGeom g;
  g.b = min(max((int)p[0], 0), B - 1);
  const float off = aligned ? 0.5f : 0.0f;
  g.x1 = p[1] * scale - off; g.y1 = p[2] * scale - off;
  float rw = p[3] * scale - off - g.x1, rh = p[4] * scale - off - g.y1;
  if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
  g.bw = rw / (float)out; g.bh = rh / (float)out;
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)out);
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)out);
  g.empty = g.gw <= 0 || g.gh <= 0;                   // non-positive size with aligned=True: the bin average is 0
  return g;
}

// bilinear corner set of one sample; returns false if the sample lies outside [-1, n] (contributes 0)
__device__ __forceinline__ bool bilinear(float y, float x, int H, int W, int& ylo, int& yhi, int& xlo, int& xhi, float& w1,
                                         float& w2, float& w3, float& w4) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return false;
  y = fmaxf(y, 0.0f); x = fmaxf(x, 0.0f);
  ylo = (int)y; xlo = (int)x;
  if (ylo >= H - 1) { yhi = ylo = H - 1; y = (float)ylo; } else { yhi = ylo + 1; }
  if (xlo >= W - 1) { xhi = xlo = W - 1; x = (float)xlo; } else { xhi = xlo + 1; }
  const float ly = y - (float)ylo, lx = x - (float)xlo, hy = 1.0f - ly, hx = 1.0f - lx;
  w1 = hy * hx; w2 = hy * lx; w3 = ly * hx; w4 = ly * lx;
  return true;
}

// grid (out*out, R); block = 64 * k threads, each thread owns float4 channel groups tid, tid + blockDim, ...
__global__ void roi_align_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ outp,
                                     int B, int H, int W, int C, int out, float scale, int sampling_ratio, int aligned) {
  const int r = blockIdx.y, bin = blockIdx.x, ph = bin / out, pw = bin - ph * out;
  const RoiGeom g = roi_geom(rois, r, scale, out, sampling_ratio, aligned, B);
  const float* fb = feat + (size_t)g.b * H * W * C;
  float* dst = outp + ((size_t)r * out * out + bin) * C;
  const float inv = g.empty ? 0.0f : 1.0f / (float)(g.gh * g.gw);
  for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!g.empty) {
      for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.y1 + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
          const float x = g.x1 + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
          int ylo, yhi, xlo, xhi;
          float w1, w2, w3, w4;
          if (!bilinear(y, x, H, W, ylo, yhi, xlo, xhi, w1, w2, w3, w4)) continue;
          const float4 a = *reinterpret_cast<const float4*>(fb + ((size_t)ylo * W + xlo) * C + c);
          const float4 b4 = *reinterpret_cast<const float4*>(fb + ((size_t)ylo * W + xhi) * C + c);
          const float4 c4 = *reinterpret_cast<const float4*>(fb + ((size_t)yhi * W + xlo) * C + c);
          const float4 d = *reinterpret_cast<const float4*>(fb + ((size_t)yhi * W + xhi) * C + c);
          acc.x += w1 * a.x + w2 * b4.x + w3 * c4.x + w4 * d.x;
          acc.y += w1 * a.y + w2 * b4.y + w3 * c4.y + w4 * d.y;
          acc.z += w1 * a.z + w2 * b4.z + w3 * c4.z + w4 * d.z;
          acc.w += w1 * a.w + w2 * b4.w + w3 * c4.w + w4 * d.w;
        }
      }
    }
    *reinterpret_cast<float4*>(dst + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

// Backward as a GATHER: grid (C / 8, B), 1024 threads.  Bin averaging + bilinear sampling is separable, so the gradient a
// RoI sends to pixel (py, px) is  sum_{ph,pw} Ay[ph][py] * Ax[pw][px] * dout[r][ph][pw][c]  with the 1-D tables
// Ay[ph][y] = (1/gh) * sum over the bin's samples of their bilinear weight on row y (Ax alike).  Every thread OWNS a fixed
// set of pixels (t, t + 1024, ...) x 8 channels in registers and walks the image's RoIs in batches (`bsz`: as many as fit the LDS, <= 32) whose
// tables and dout slices are staged in LDS: no atomics at all (a scatter with LDS float atomics took 5.7 ms for 1024
// RoIs -- the proposals of an object overlap the same pixels; global float atomics ~1 s), fixed summation order,
// deterministic.
constexpr int RB_NT = 1024, RB_BATCH_MAX = 32, RB_CH = 8, RB_MAXPT = 8;        // pixels per thread <= 8: maps up to 8192 pixels

template <int MAXPT>   // pixels per thread (accumulators: MAXPT x 8 registers): 4 for maps up to 4096 pixels, else 8
__global__ __launch_bounds__(RB_NT) void roi_align_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ rois,
                                                              float* __restrict__ dfeat, int B, int H, int W, int C, int R,
                                                              int out, float scale, int sampling_ratio, int aligned, int bsz) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int c0 = blockIdx.x * RB_CH, b = blockIdx.y, tid = threadIdx.x;
  const int npix = H * W, nb = out * out;
  float* Ay = sm;                                   // [bsz][out][H]
  float* Ax = Ay + bsz * out * H;                   // [bsz][out][W]
  float* Dd = sm + ((bsz * out * (H + W) + 3) & ~3);   // [bsz][nb][RB_CH], float4-staged: the tables are padded to 16 bytes
  int* span = reinterpret_cast<int*>(Dd + bsz * nb * RB_CH);           // [bsz][4] = ylo, yhi, xlo, xhi (inclusive)
  // per (RoI, pixel row / column): the first and last bin with a non-zero weight there, packed lo | hi << 16 -- a pixel
  // touches 1-3 bins per axis, not all `out` of them (filled from the finished tables, one thread per position)
  int* brange = span + bsz * 4;                     // [bsz][H + W]
  int* cnt_s = brange + bsz * (H + W);              // [4] counters (all LDS lives in the dynamic region)
  int& nlist_s = cnt_s[0];
  int* list_s = cnt_s + 4;                          // [R] RoIs of this image, RoI order

  float acc[MAXPT][RB_CH];
#pragma unroll
  for (int k = 0; k < MAXPT; ++k)
#pragma unroll
    for (int j = 0; j < RB_CH; ++j) acc[k][j] = 0.0f;

  // the image's RoIs with a non-empty sample grid, compacted once (RoI order) into LDS
  if (tid == 0) nlist_s = 0;
  __syncthreads();
  for (int r0 = 0; r0 < R; r0 += RB_NT) {
    const int r = r0 + tid;
    bool mine = false;
    if (r < R) {
      const RoiGeom g = roi_geom(rois, r, scale, out, sampling_ratio, aligned, B);
      mine = g.b == b && !g.empty;
    }
    const unsigned long long m = __ballot(mine);
    // waves append in wave order: wave w waits for its turn through a running counter (16 waves, trivial cost)
    const int wave = tid >> 6, lane = tid & 63;
    for (int w = 0; w < RB_NT / 64; ++w) {
      if (w == wave) {
        const int base = nlist_s;
        if (mine) list_s[base + __popcll(m & ((1ull << lane) - 1ull))] = r;
        if (lane == 0) nlist_s = base + __popcll(m);
      }
      __syncthreads();
    }
  }
  const int ntot = nlist_s;

  for (int l0 = 0; l0 < ntot; l0 += bsz) {
    const int nl = min(bsz, ntot - l0);
    __syncthreads();                                // previous batch fully consumed
    for (int i = tid; i < bsz * out * (H + W); i += RB_NT) sm[i] = 0.0f;      // Ay and Ax are contiguous
    if (tid < bsz * 4) span[tid] = (tid & 1) ? -1 : (1 << 30);                  // lo = +inf, hi = -1
    __syncthreads();
    // 1-D tables: one thread per (RoI of the batch, bin, axis) walks that bin's samples sequentially and widens the
    // RoI's non-zero span on that axis
    if (tid < nl * out * 2) {
      const int j = tid / (out * 2), rem = tid - j * out * 2, axis = rem / out, p = rem - axis * out;
      const RoiGeom g = roi_geom(rois, list_s[l0 + j], scale, out, sampling_ratio, aligned, B);
      const int n = axis == 0 ? H : W, gs = axis == 0 ? g.gh : g.gw;
      const float start = axis == 0 ? g.y1 : g.x1, bsz = axis == 0 ? g.bh : g.bw;
      float* tab = (axis == 0 ? Ay + (j * out + p) * H : Ax + (j * out + p) * W);
      const float inv = 1.0f / (float)gs;
      int lo_all = 1 << 30, hi_all = -1;
      for (int i = 0; i < gs; ++i) {
        float c = start + (float)p * bsz + ((float)i + 0.5f) * bsz / (float)gs;
        if (c < -1.0f || c > (float)n) continue;     // this sample contributes nothing (on this axis)
        c = fmaxf(c, 0.0f);
        int lo = (int)c, hi;
        if (lo >= n - 1) { hi = lo = n - 1; c = (float)lo; } else { hi = lo + 1; }
        const float l = c - (float)lo;
        tab[lo] += (1.0f - l) * inv;
        tab[hi] += l * inv;
        lo_all = min(lo_all, lo); hi_all = max(hi_all, hi);
      }
      atomicMin(&span[j * 4 + axis * 2], lo_all);
      atomicMax(&span[j * 4 + axis * 2 + 1], hi_all);
    }
    for (int i = tid; i < nl * nb * (RB_CH / 4); i += RB_NT) {          // dout slices of the batch
      const int j = i / (nb * (RB_CH / 4)), rem = i - j * nb * (RB_CH / 4), bin = rem / (RB_CH / 4), q = rem - bin * (RB_CH / 4);
      *reinterpret_cast<float4*>(Dd + (j * nb + bin) * RB_CH + q * 4) =
          *reinterpret_cast<const float4*>(dout + ((size_t)list_s[l0 + j] * nb + bin) * C + c0 + q * 4);
    }
    __syncthreads();
    // bin ranges: one thread per (RoI, pixel row or column) scans that position's column of the finished table
    for (int i = tid; i < bsz * (H + W); i += RB_NT) {
      const int j = i / (H + W), q = i - j * (H + W);
      const float* col = q < H ? Ay + j * out * H + q : Ax + j * out * W + (q - H);
      const int stride = q < H ? H : W;
      int lo = 0x7fff, hi = 0;                                          // empty: lo > hi
      if (j < nl)
        for (int p = 0; p < out; ++p)
          if (col[p * stride] != 0.0f) { lo = min(lo, p); hi = p; }
      brange[i] = lo | (hi << 16);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXPT; ++k) {
      const int pix = tid + k * RB_NT;
      if (pix >= npix) break;
      const int py = pix / W, px = pix - py * W;
      for (int j = 0; j < nl; ++j) {
        if (py < span[j * 4] || py > span[j * 4 + 1] || px < span[j * 4 + 2] || px > span[j * 4 + 3]) continue;
        const float* ay = Ay + j * out * H + py;
        const float* ax = Ax + j * out * W + px;
        const float* d = Dd + j * nb * RB_CH;
        const int ry = brange[j * (H + W) + py], rx = brange[j * (H + W) + H + px];
        const int ph1 = ry >> 16, pw0 = rx & 0xffff, pw1 = rx >> 16;
        for (int ph = ry & 0xffff; ph <= ph1; ++ph) {
          const float wy = ay[ph * H];
          if (wy == 0.0f) continue;
          for (int pw = pw0; pw <= pw1; ++pw) {
            const float w = wy * ax[pw * W];
            if (w == 0.0f) continue;
            const float4 d0 = *reinterpret_cast<const float4*>(d + (ph * out + pw) * RB_CH);
            const float4 d1 = *reinterpret_cast<const float4*>(d + (ph * out + pw) * RB_CH + 4);
            acc[k][0] = fmaf(w, d0.x, acc[k][0]); acc[k][1] = fmaf(w, d0.y, acc[k][1]);
            acc[k][2] = fmaf(w, d0.z, acc[k][2]); acc[k][3] = fmaf(w, d0.w, acc[k][3]);
            acc[k][4] = fmaf(w, d1.x, acc[k][4]); acc[k][5] = fmaf(w, d1.y, acc[k][5]);
            acc[k][6] = fmaf(w, d1.z, acc[k][6]); acc[k][7] = fmaf(w, d1.w, acc[k][7]);
          }
        }
      }
    }
  }
  float* dst = dfeat + (size_t)b * npix * C + c0;
#pragma unroll
  for (int k = 0; k < MAXPT; ++k) {
    const int pix = tid + k * RB_NT;
    if (pix >= npix) break;
    *reinterpret_cast<float4*>(dst + (size_t)pix * C) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
    *reinterpret_cast<float4*>(dst + (size_t)pix * C + 4) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
  }
}

int roi_block(int C) {
  const int t = as_round_up(as_ceil_div(C, 4), 64);
  return t > 256 ? 256 : t;
}

}  // namespace

extern "C" int as_roi_align_fwd(const float* feat, const float* rois, float* out, int B, int H, int W, int C, int R,
                                int out_size, float spatial_scale, int sampling_ratio, int aligned, as_stream_t stream) {
  AS_REQUIRE(feat && out && (rois || R == 0), AS_E_BADARG, "as_roi_align_fwd: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && R >= 0 && out_size > 0, AS_E_BADARG, "as_roi_align_fwd: bad sizes");
  AS_REQUIRE(C % 4 == 0, AS_E_UNSUPPORTED, "as_roi_align_fwd: C=%d must be a multiple of 4", C);
  if (R == 0) return AS_OK;
  hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(out_size * out_size, R), dim3(roi_block(C)), 0, (hipStream_t)stream, feat,
                     rois, out, B, H, W, C, out_size, spatial_scale, sampling_ratio, aligned);
  AS_CHECK_LAUNCH("roi_align_fwd");
  return AS_OK;
}

extern "C" int as_roi_align_bwd(const float* dout, const float* rois, float* dfeat, int B, int H, int W, int C, int R,
                                int out_size, float spatial_scale, int sampling_ratio, int aligned, as_stream_t stream) {
  AS_REQUIRE(dfeat && (dout || R == 0) && (rois || R == 0), AS_E_BADARG, "as_roi_align_bwd: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && R >= 0 && out_size > 0, AS_E_BADARG, "as_roi_align_bwd: bad sizes");
  AS_REQUIRE(C % RB_CH == 0, AS_E_UNSUPPORTED, "as_roi_align_bwd: C=%d must be a multiple of %d", C, RB_CH);
  AS_REQUIRE(H * W <= RB_NT * RB_MAXPT, AS_E_UNSUPPORTED, "as_roi_align_bwd: a %dx%d map exceeds %d pixels", H, W, RB_NT * RB_MAXPT);
  // RoIs per batch: as many as the LDS takes (every batch costs a fixed round of barriers and a table build by
  // 2 * out threads per RoI), at most RB_BATCH_MAX and RB_NT / (2 * out) (one table thread per RoI, bin and axis)
  auto lds_of = [&](int bsz) {
    return ((((size_t)bsz * out_size * (H + W) + 3) & ~(size_t)3) + (size_t)bsz * out_size * out_size * RB_CH + (size_t)bsz * 4 + (size_t)bsz * (H + W) + 4 +
            (size_t)R) * 4;
  };
  int bsz = RB_BATCH_MAX;
  while (bsz > 1 && (lds_of(bsz) > 150 * 1024 || out_size * 2 * bsz > RB_NT)) --bsz;
  AS_REQUIRE(out_size * 2 * bsz <= RB_NT, AS_E_UNSUPPORTED, "as_roi_align_bwd: output size %d", out_size);
  const size_t lds = lds_of(bsz);
  AS_REQUIRE(lds <= 150 * 1024, AS_E_UNSUPPORTED, "as_roi_align_bwd: tables of a %dx%d map / output %d exceed LDS", H, W, out_size);
  static std::atomic<bool> attr{false};
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)roi_align_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)roi_align_bwd_kernel<RB_MAXPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  if (H * W <= RB_NT * 4)
    hipLaunchKernelGGL(roi_align_bwd_kernel<4>, dim3(C / RB_CH, B), dim3(RB_NT), lds, (hipStream_t)stream, dout, rois, dfeat, B, H,
                       W, C, R, out_size, spatial_scale, sampling_ratio, aligned, bsz);
  else
    hipLaunchKernelGGL(roi_align_bwd_kernel<RB_MAXPT>, dim3(C / RB_CH, B), dim3(RB_NT), lds, (hipStream_t)stream, dout, rois, dfeat,
                       B, H, W, C, R, out_size, spatial_scale, sampling_ratio, aligned, bsz);
  AS_CHECK_LAUNCH("roi_align_bwd");
  return AS_OK;
}
