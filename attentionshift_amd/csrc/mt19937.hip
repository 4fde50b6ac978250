// torch's CPU generator on the device: the draws of the reference-RNG mode without a host round trip.
//
// The reference samples its seed points and mask points with torch.randint / torch.randperm from the GLOBAL CPU generator
// (stdroi:343-371 sample_point_grid, :433-461 get_mask_points_single_instance); how many engine words a call consumes
// depends on candidate COUNTS that only exist on the device (n_draw = ceil(n / (n / k)) words per randint, n - 1 per
// randperm(n)).  Reading the counts back to draw on the host costs three blocking syncs per image.  Here the engine
// state (624 words + position, attentionshift_amd/mt19937.py) is uploaded once per call, ONE workgroup advances it exactly as
// torch's mt19937 would -- the refill is the textbook recurrence, data-parallel in three dependent levels -- and the
// final state goes back into torch at the end of the call: identical draws, identical generator afterwards
// (tests/test_mt19937.py pins the arithmetic to torch on the CPU, tests/test_gpu_kernels.py the kernels to it).
#include "common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

constexpr int MT_NT = 256;               // one workgroup of four waves: a refill is three 227-wide dependent levels

// The engine in LDS, driven by ALL threads of the workgroup with uniform control flow.  `st` points at the current state
// array, `alt` at the spare one: next_state() writes the new block into `alt` (no read-after-write hazards between
// neighbours, three barriers per refill) and swaps the two.
struct MtBlock {
  unsigned* st;
  unsigned* alt;
  int left, next, tid;

  __device__ __forceinline__ static unsigned twist(unsigned u, unsigned v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
  }
  // next_state(): new[i] = old[i + M] ^ twist(old[i], old[i + 1]) for i < N - M = 227 (level 1: old operands only);
  // new[i] = new[i - 227] ^ twist(old[i], old[i + 1]) for 227 <= i < 454 (level 2: needs level 1) and for 454 <= i < 623
  // (level 3: needs level 2); new[623] = new[396] ^ twist(old[623], new[0]) rides in level 3.
  __device__ void refill() {
    if (tid < MT_N - MT_M) alt[tid] = st[tid + MT_M] ^ twist(st[tid], st[tid + 1]);
    __syncthreads();
    if (tid < MT_N - MT_M) alt[tid + 227] = alt[tid] ^ twist(st[tid + 227], st[tid + 228]);
    __syncthreads();
    if (tid < 169) alt[tid + 454] = alt[tid + 227] ^ twist(st[tid + 454], st[tid + 455]);
    if (tid == 255) alt[MT_N - 1] = alt[MT_M - 1] ^ twist(st[MT_N - 1], alt[0]);
    __syncthreads();
    unsigned* t = st; st = alt; alt = t;
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
  // up to 64 consecutive words, one per thread (thread t < c gets word t of the run); returns c = min(want, 64, ready())
  __device__ __forceinline__ int take(int want, unsigned& word) {
    const bool fresh = left == 1;
    const int avail = ready();
    int c = want < 64 ? want : 64;
    c = c < avail ? c : avail;
    word = tid < c ? temper(st[next + tid]) : 0u;
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

__device__ __forceinline__ void mt_load(MtBlock& w, unsigned* lds, const int* state, int tid) {
  for (int i = tid; i < MT_N; i += MT_NT) lds[i] = (unsigned)state[i];
  __syncthreads();
  w.st = lds;
  w.alt = lds + MT_N + 8;
  w.left = state[MT_N];
  w.next = state[MT_N + 1];
  w.tid = tid;
}
__device__ __forceinline__ void mt_store(const MtBlock& w, int* state, int tid) {
  __syncthreads();
  for (int i = tid; i < MT_N; i += MT_NT) state[i] = (int)w.st[i];
  if (tid == 0) { state[MT_N] = w.left; state[MT_N + 1] = w.next; }
}

// S candidate sets: ranks[s][0..k) = (torch.randint(n_s, (n_draw_s,)) % n_s)[:k], n_draw_s = ceil(n_s / (n_s / k)), in set order.
// flag |= 1: a set with fewer than k candidates (stdroi:354-364 refill branches) or a range torch would draw 64-bit words for.
__global__ __launch_bounds__(MT_NT) void mt_sample_kernel(int* __restrict__ state, const int* __restrict__ counts, int S, int k,
                                                          int* __restrict__ ranks, int* __restrict__ flag) {
  __shared__ unsigned lds[2 * MT_N + 8];
  const int tid = threadIdx.x;
  MtBlock w;
  mt_load(w, lds, state, tid);
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
      if (tid < c && d + tid < k) ranks[s * k + d + tid] = (int)(word % (unsigned)n);
      d += c;
    }
  }
  mt_store(w, state, tid);
  if (tid == 0) flag[0] = bad;
}

// G objects: ranks[g][0..k) = torch.randperm(n_g)[:k] with n_g = counts[g][0] + counts[g][1] (forward Fisher-Yates: the first k
// entries are final after k swaps; the remaining n_g - 1 - k words are skipped).  flag |= 1: n_g < k (the reference's
// fill-in / empty branches, stdroi:449-455) or n_g beyond the simple-loop range of randperm.
__global__ __launch_bounds__(MT_NT) void mt_perm_kernel(int* __restrict__ state, const int* __restrict__ counts2, int G, int k,
                                                        int* __restrict__ ranks, int* __restrict__ flag) {
  __shared__ unsigned lds[2 * MT_N + 8];
  __shared__ int mpos[64], mval[64];                          // positions touched by the first k swaps (<= 2k) and their values
  __shared__ unsigned wbox;
  const int tid = threadIdx.x;
  MtBlock w;
  mt_load(w, lds, state, tid);
  int bad = 0;
  for (int g = 0; g < G; ++g) {
    const long long n = (long long)counts2[2 * g] + counts2[2 * g + 1];
    if (n < k || n >= (long long)(0xffffffffu / 20)) { bad = 1; continue; }
    const int steps = (int)((n - 1) < (long long)k ? (n - 1) : (long long)k);      // swaps that decide the first k entries
    int nm = 0;
    for (int i = 0; i < steps; ++i) {                        // sequential by nature; every thread runs it redundantly (uniform)
      unsigned word;
      (void)w.take(1, word);
      if (tid == 0) wbox = word;
      __syncthreads();
      word = wbox;
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
      __syncthreads();
      if (tid == 0) {
        if (new_a) mpos[sa] = pa;
        mval[sa] = vb;
        if (pb != pa) {
          if (new_b) mpos[sb] = pb;
          mval[sb] = va;
        }
      }
      nm += (new_a ? 1 : 0) + (new_b ? 1 : 0);
      __syncthreads();
    }
    if (tid < k) {
      int v = tid;
      for (int t = 0; t < nm; ++t)
        if (mpos[t] == tid) v = mval[t];
      ranks[g * k + tid] = v;
    }
    __syncthreads();
    w.skip((n - 1) - steps);
  }
  mt_store(w, state, tid);
  if (tid == 0) flag[0] = bad;
}

}  // namespace

extern "C" int as_mt_sample_ranks(int32_t* state, const int32_t* counts, int32_t* ranks, int32_t* flag, int S, int K,
                                  as_stream_t stream) {
  AS_REQUIRE(state && counts && ranks && flag, AS_E_BADARG, "as_mt_sample_ranks: null pointer");
  AS_REQUIRE(S > 0 && K > 0 && K <= 64, AS_E_BADARG, "as_mt_sample_ranks: bad sizes S=%d K=%d", S, K);
  hipLaunchKernelGGL(mt_sample_kernel, dim3(1), dim3(MT_NT), 0, (hipStream_t)stream, state, counts, S, K, ranks, flag);
  AS_CHECK_LAUNCH("mt_sample");
  return AS_OK;
}

extern "C" int as_mt_perm_ranks(int32_t* state, const int32_t* counts2, int32_t* ranks, int32_t* flag, int G, int K,
                                as_stream_t stream) {
  AS_REQUIRE(state && counts2 && ranks && flag, AS_E_BADARG, "as_mt_perm_ranks: null pointer");
  AS_REQUIRE(G > 0 && K > 0 && K <= 32, AS_E_BADARG, "as_mt_perm_ranks: bad sizes G=%d K=%d", G, K);
  hipLaunchKernelGGL(mt_perm_kernel, dim3(1), dim3(MT_NT), 0, (hipStream_t)stream, state, counts2, G, K, ranks, flag);
  AS_CHECK_LAUNCH("mt_perm");
  return AS_OK;
}
