// Shared device/host helpers for libattnshift_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/attnshift.h"

// ---------------------------------------------------------------------------------------------
// error plumbing: never throw / exit across the C boundary
// ---------------------------------------------------------------------------------------------
void as_set_error(const char* fmt, ...);

#define AS_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      as_set_error(__VA_ARGS__);               \
      return (code);                           \
    }                                          \
  } while (0)

#define AS_CHECK_LAUNCH(what)                                                        \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      as_set_error("%s: launch failed: %s", (what), hipGetErrorString(e_));          \
      return AS_E_LAUNCH;                                                            \
    }                                                                                \
  } while (0)

static inline int as_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int as_round_up(int a, int b) { return as_ceil_div(a, b) * b; }

// ---------------------------------------------------------------------------------------------
// Fork / join onto a helper stream inside one C call: two independent launch chains of a backward pass run side by side
// (a chain of one-workgroup-per-CU split-K products leaves most of every CU free; two half-empty last rounds fill each
// other's slots).  The caller's stream forks with an event, the helper chain runs on the side stream, and the caller's
// stream waits for the helper's last launch before the call returns -- so the caller sees ordinary stream order.
// One set of stream + events per host thread, device and slot (nested users take different slots).  AS_BWD_SERIAL=1 (read
// once) keeps everything on the caller's stream.
// ---------------------------------------------------------------------------------------------
struct AsSide {
  // one stream + three events PER DEVICE (a thread alternating between devices reuses each device's set instead of
  // re-creating -- and leaking -- one on every switch); a set whose creation fails half-way is destroyed and stays off
  static constexpr int kMaxDev = 16;
  struct Slot {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, mid = nullptr;
    int state = 0;                                      // 0 = not tried, 1 = ready, -1 = failed (do not retry)
  };
  Slot slots[kMaxDev];
  hipStream_t st = nullptr;                             // the CURRENT device's set (valid while ok)
  hipEvent_t fork = nullptr, join = nullptr, mid = nullptr;
  bool ok = false;
  void init() {
    int d = -1;
    ok = false;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) return;
    Slot& sl = slots[d];
    if (sl.state == 0) {
      const bool made = hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&sl.fork, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&sl.join, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&sl.mid, hipEventDisableTiming) == hipSuccess;
      if (!made) {
        if (sl.mid) (void)hipEventDestroy(sl.mid);
        if (sl.join) (void)hipEventDestroy(sl.join);
        if (sl.fork) (void)hipEventDestroy(sl.fork);
        if (sl.st) (void)hipStreamDestroy(sl.st);
        sl = Slot();
        (void)hipGetLastError();                        // the caller falls back to its own stream: not its error
      }
      sl.state = made ? 1 : -1;
    }
    if (sl.state != 1) return;
    st = sl.st; fork = sl.fork; join = sl.join; mid = sl.mid;
    ok = true;
  }
  // thread exit (the instances are thread_local): give the streams and events back.  Errors are ignored -- at process exit
  // the HIP runtime may already be gone.
  ~AsSide() {
    for (Slot& sl : slots) {
      if (sl.state != 1) continue;
      (void)hipEventDestroy(sl.mid);
      (void)hipEventDestroy(sl.join);
      (void)hipEventDestroy(sl.fork);
      (void)hipStreamDestroy(sl.st);
      sl = Slot();
    }
    (void)hipGetLastError();
  }
};
static inline bool as_side_serial() {
  static const bool serial = getenv("AS_BWD_SERIAL") != nullptr;
  return serial;
}
// -> the stream the helper chain should use: the side stream behind an event of `s`, or `s` itself (serial / no stream)
static inline hipStream_t as_side_fork(AsSide& sd, hipStream_t s) {
  if (as_side_serial()) return s;
  sd.init();
  if (sd.ok && hipEventRecord(sd.fork, s) == hipSuccess && hipStreamWaitEvent(sd.st, sd.fork, 0) == hipSuccess) return sd.st;
  return s;
}
// `to` waits for everything queued on `from` so far (no-op when they are the same stream)
static inline void as_side_wait(AsSide& sd, hipEvent_t ev, hipStream_t from, hipStream_t to) {
  if (from == to) return;
  if (hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess)
    (void)hipStreamSynchronize(from);                   // (a lost device; keeps the order safe)
}
static inline void as_side_join(AsSide& sd, hipStream_t sq, hipStream_t s) { as_side_wait(sd, sd.join, sq, s); }
#ifdef __HIPCC__
__device__ __forceinline__ int as_ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
#endif

// ---- internal (not part of the C ABI): weight gradients without transposed activation copies, csrc/gemm_tn.hip ----
bool as_tn_applies(int M, int Nout, int K);                          // 128-aligned feature counts, 32-bit row offsets
size_t as_tn_workspace_bytes(int M, int Nout, int K);                // fp32 partials [S][Nout][K]
size_t as_tn_colsum_workspace_bytes(int C);
// db (nullable): the bias gradient falls out of the same pass; db_part = [<= 32][Nout] floats of scratch
int as_tn_dw(const void* dy, const void* x, void* dW, float* db, float* db_part, int M, int Nout, int K, int dw_f32, void* ws,
             size_t ws_bytes, hipStream_t s);
int as_tn_colsum(const void* g, float* out, float* part, int R, int C, hipStream_t s);   // bf16 [R, C] -> fp32 [C]

// ---------------------------------------------------------------------------------------------
// MFMA fragments.  One "k16 step" of a 32x32 tile: lane l = (i = l & 31, half = l >> 5) holds the 8
// consecutive k elements k0 + 8*half .. +7 of row i (A) / column i (B).  C/D: col = l & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) for accumulator register r (guide section 3).
//   bf16 : one v_mfma_f32_32x32x16_bf16
//   f32  : eight v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, k-ordered), MFMA t contracts the
//          k pair {element t of half 0, element t of half 1}
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

#ifdef __HIPCC__
template <typename T> struct Frag;

template <> struct Frag<__bf16> {
  bf16x8 v;
  __device__ __forceinline__ void load16B(const __bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = (__bf16)0.0f;
  }
  __device__ __forceinline__ void set(int t, float x) { v[t] = (__bf16)x; }
};
template <> struct Frag<float> {
  float v[8];
  __device__ __forceinline__ void load16B(const float* p) {   // 32 bytes actually: 8 floats
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = 0.0f;
  }
  __device__ __forceinline__ void set(int t, float x) { v[t] = x; }
};

__device__ __forceinline__ f32x16 mma32(const Frag<__bf16>& a, const Frag<__bf16>& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32(const Frag<float>& a, const Frag<float>& b, f32x16 c) {
#pragma unroll
  for (int t = 0; t < 8; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], b.v[t], c, 0, 0, 0);
  return c;
}

// Q is stored FRAGMENT-MAJOR so that the row-per-lane 16-byte operand loads of the MFMAs are fully coalesced:
// inside every 32-row x 64-d tile, element (r, d) lives at ((d/16)*64 + r + 32*((d%16)/8))*8 + d%8, i.e. the 8
// elements lane (r, half) needs for k16-step ks are contiguous and the 64 lanes of a wave read 1 KiB contiguous.
// (q workspace of as_qkv_fwd; consumers: sdpa.hip, rollout.hip.)
__device__ __forceinline__ size_t qf_frag(size_t bh, int Npad, int row, int ks, int half) {     // element index
  return ((((bh * (size_t)(Npad >> 5) + (size_t)(row >> 5)) * 4 + ks) * 64) + (row & 31) + 32 * half) * 8;
}
__device__ __forceinline__ size_t qf_elem(size_t bh, int Npad, int row, int d) {
  return qf_frag(bh, Npad, row, d >> 4, (d >> 3) & 1) + (d & 7);
}

// q is stored PRE-SCALED: q' = (x Wq^T + bq) * log2(e) / sqrt(64), rounded once from the fp32 accumulator (QKV epilogue,
// gemm.hip).  Every consumer then gets base-2 logits straight from the MFMA, s' = q' . k = log2(e) * (q . k) / 8:
// softmax weights are exp2(s' - lse * log2 e) with no multiply per score (sdpa.hip, rollout.hip, sdpa_bwd.hip), and
// d/dk carries ln 2 instead of 1/8 (sdpa_bwd.hip).
#define AS_LOG2E 1.44269504088896340736f
#define AS_LN2 0.69314718055994530942f
#define AS_QSCALE (0.125f * AS_LOG2E)

// accumulator register r of a lane in half `half` -> row inside the 32x32 tile
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<__bf16>(__bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float x) { return (__bf16)x; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#endif
