// torch's CPU generator on the device: the draws of the reference-RNG mode without a host round trip.
//
// The reference samples its seed points and mask points with torch.randint / torch.randperm from the GLOBAL CPU generator
// (stdroi:343-371 sample_point_grid, :433-461 get_mask_points_single_instance); how many engine words a call consumes
// depends on candidate COUNTS that only exist on the device (n_draw = ceil(n / (n / k)) words per randint, n - 1 per
// randperm(n)).  Reading the counts back to draw on the host costs three blocking syncs per image.  Here the engine
// state (624 words + position, attentionshift_amd/mt19937.py) is uploaded once per call, ONE wave advances it exactly as
// torch's mt19937 would -- the refill is the textbook recurrence, data-parallel in three dependent spans -- and the
// final state goes back into torch at the end of the call: identical draws, identical generator afterwards
// (tests/test_mt19937.py pins the arithmetic to torch on the CPU, tests/test_gpu_kernels.py the kernels to it).
#include "common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

struct MtWave {                         // one wave; state words in LDS, position in (uniform) registers
  unsigned* st;
  int left, next, lane;

  __device__ __forceinline__ static unsigned twist(unsigned u, unsigned v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
  }
  // next_state(): s[i] = s[i + M] ^ twist(s[i], s[i + 1]) for i < N - M (operands all OLD), then s[i] = s[i + M - N] ^ ... for
  // i < N - 1 (first operand NEW: distance 227, so chunks of 64 in increasing order are independent inside a chunk), then the
  // last word with the new s[0].  A wave executes an LDS load before the store that follows it for ALL its lanes, so
  // "read old neighbours, then write" needs no extra barrier inside a chunk.
  __device__ void refill() {
    for (int i0 = 0; i0 < MT_N - MT_M; i0 += 64) {
      const int i = i0 + lane;
      unsigned v = 0;
      if (i < MT_N - MT_M) v = st[i + MT_M] ^ twist(st[i], st[i + 1]);
      __builtin_amdgcn_wave_barrier();
      if (i < MT_N - MT_M) st[i] = v;
      __builtin_amdgcn_wave_barrier();
    }
    for (int i0 = MT_N - MT_M; i0 < MT_N - 1; i0 += 64) {
      const int i = i0 + lane;
      unsigned v = 0;
      if (i < MT_N - 1) v = st[i + MT_M - MT_N] ^ twist(st[i], st[i + 1]);
      __builtin_amdgcn_wave_barrier();
      if (i < MT_N - 1) st[i] = v;
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) st[MT_N - 1] = st[MT_M - 1] ^ twist(st[MT_N - 1], st[0]);
    __builtin_amdgcn_wave_barrier();
    left = MT_N;
    next = 0;
  }
  __device__ __forceinline__ static unsigned temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  // words the engine can hand out before the next refill, after making sure it is >= 1 (torch: `if (--left == 0)
  // next_state()`: the word that finds left == 1 triggers the refill and is the first of the new block, whose `left` stays N)
  __device__ __forceinline__ int ready() {
    if (left == 1) {
      refill();
      return MT_N;                      // the first word of a fresh block does not decrement `left`
    }
    return left - 1;
  }
  __device__ __forceinline__ void consumed(int c, bool fresh) {   // c words taken from a block (fresh: it was just refilled)
    next += c;
    left -= fresh ? c - 1 : c;
  }
  // up to 64 consecutive words, one per lane (lane l < c gets word l of the run); returns c = min(want, 64, ready())
  __device__ __forceinline__ int take(int want, unsigned& word) {
    const bool fresh = left == 1;
    const int avail = ready();
    int c = want < 64 ? want : 64;
    c = c < avail ? c : avail;
    word = lane < c ? temper(st[next + lane]) : 0u;
    consumed(c, fresh);
    return c;
  }
  __device__ void skip(long long m) {
    while (m > 0) {
      const bool fresh = left == 1;
      const int avail = ready();
      const int c = m < (long long)avail ? (int)m : avail;
      consumed(c, fresh);
      m -= c;
    }
  }
};

__device__ __forceinline__ void mt_load(MtWave& w, unsigned* lds, const int* state, int lane) {
  for (int i = lane; i < MT_N; i += 64) lds[i] = (unsigned)state[i];
  __builtin_amdgcn_wave_barrier();
  w.st = lds;
  w.left = state[MT_N];
  w.next = state[MT_N + 1];
  w.lane = lane;
}
__device__ __forceinline__ void mt_store(const MtWave& w, int* state, int lane) {
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < MT_N; i += 64) state[i] = (int)w.st[i];
  if (lane == 0) { state[MT_N] = w.left; state[MT_N + 1] = w.next; }
}

// S candidate sets: ranks[s][0..k) = (torch.randint(n_s, (n_draw_s,)) % n_s)[:k], n_draw_s = ceil(n_s / (n_s / k)), in set order.
// flag |= 1: a set with fewer than k candidates (stdroi:354-364 refill branches) or a range torch would draw 64-bit words for.
__global__ __launch_bounds__(64) void mt_sample_kernel(int* __restrict__ state, const int* __restrict__ counts, int S, int k,
                                                       int* __restrict__ ranks, int* __restrict__ flag) {
  __shared__ unsigned lds[MT_N];
  const int lane = threadIdx.x;
  MtWave w;
  mt_load(w, lds, state, lane);
  int bad = 0;
  for (int s = 0; s < S; ++s) {
    const int n = counts[s];
    if (n < k || n >= (1 << 28)) { bad = 1; continue; }     // (host path; nothing is consumed for it here)
    const int step = n / k;
    const int n_draw = (n + step - 1) / step;                // len(range(0, n, n // k))
    int d = 0;
    while (d < n_draw) {
      unsigned word;
      const int c = w.take(n_draw - d, word);
      if (lane < c && d + lane < k) ranks[s * k + d + lane] = (int)(word % (unsigned)n);
      d += c;
    }
  }
  mt_store(w, state, lane);
  if (lane == 0) flag[0] = bad;
}

// G objects: ranks[g][0..k) = torch.randperm(n_g)[:k] with n_g = counts[g][0] + counts[g][1] (forward Fisher-Yates: the first k
// entries are final after k swaps; the remaining n_g - 1 - k words are skipped).  flag |= 1: n_g < k (the reference's
// fill-in / empty branches, stdroi:449-455) or n_g beyond the simple-loop range of randperm.
__global__ __launch_bounds__(64) void mt_perm_kernel(int* __restrict__ state, const int* __restrict__ counts2, int G, int k,
                                                     int* __restrict__ ranks, int* __restrict__ flag) {
  __shared__ unsigned lds[MT_N];
  __shared__ int mpos[64], mval[64];                          // positions touched by the first k swaps (<= 2k) and their values
  const int lane = threadIdx.x;
  MtWave w;
  mt_load(w, lds, state, lane);
  int bad = 0;
  for (int g = 0; g < G; ++g) {
    const long long n = (long long)counts2[2 * g] + counts2[2 * g + 1];
    if (n < k || n >= (long long)(0xffffffffu / 20)) { bad = 1; continue; }
    const int steps = (int)((n - 1) < (long long)k ? (n - 1) : (long long)k);      // swaps that decide the first k entries
    int nm = 0;
    for (int i = 0; i < steps; ++i) {                        // sequential by nature; every lane runs it redundantly (uniform)
      unsigned word;
      (void)w.take(1, word);
      word = __shfl(word, 0);
      const int z = (int)(word % (unsigned)(n - i));
      const int pa = i, pb = z + i;
      int ia = -1, ib = -1;
      for (int t = 0; t < nm; ++t) {
        if (mpos[t] == pa) ia = t;
        if (mpos[t] == pb) ib = t;
      }
      const int va = ia >= 0 ? mval[ia] : pa, vb = ib >= 0 ? mval[ib] : pb;
      const bool new_a = ia < 0, new_b = pb != pa && ib < 0;
      const int sa = new_a ? nm : ia, sb = new_b ? nm + (new_a ? 1 : 0) : ib;      // slots of the two positions
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        if (new_a) mpos[sa] = pa;
        mval[sa] = vb;
        if (pb != pa) {
          if (new_b) mpos[sb] = pb;
          mval[sb] = va;
        }
      }
      nm += (new_a ? 1 : 0) + (new_b ? 1 : 0);
      __builtin_amdgcn_wave_barrier();
    }
    if (lane < k) {
      int v = lane;
      for (int t = 0; t < nm; ++t)
        if (mpos[t] == lane) v = mval[t];
      ranks[g * k + lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    w.skip((n - 1) - steps);
  }
  mt_store(w, state, lane);
  if (lane == 0) flag[0] = bad;
}

}  // namespace

extern "C" int as_mt_sample_ranks(int32_t* state, const int32_t* counts, int32_t* ranks, int32_t* flag, int S, int K,
                                  as_stream_t stream) {
  AS_REQUIRE(state && counts && ranks && flag, AS_E_BADARG, "as_mt_sample_ranks: null pointer");
  AS_REQUIRE(S > 0 && K > 0 && K <= 64, AS_E_BADARG, "as_mt_sample_ranks: bad sizes S=%d K=%d", S, K);
  hipLaunchKernelGGL(mt_sample_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, counts, S, K, ranks, flag);
  AS_CHECK_LAUNCH("mt_sample");
  return AS_OK;
}

extern "C" int as_mt_perm_ranks(int32_t* state, const int32_t* counts2, int32_t* ranks, int32_t* flag, int G, int K,
                                as_stream_t stream) {
  AS_REQUIRE(state && counts2 && ranks && flag, AS_E_BADARG, "as_mt_perm_ranks: null pointer");
  AS_REQUIRE(G > 0 && K > 0 && K <= 32, AS_E_BADARG, "as_mt_perm_ranks: bad sizes G=%d K=%d", G, K);
  hipLaunchKernelGGL(mt_perm_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, counts2, G, K, ranks, flag);
  AS_CHECK_LAUNCH("mt_perm");
  return AS_OK;
}
