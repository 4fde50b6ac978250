"""torch's CPU generator (mt19937, aten/src/ATen/core/MT19937RNGEngine.h) as data: the bridge that lets the
reference-RNG mode make its draws ON THE DEVICE (csrc/mt19937.hip) and still consume exactly the stream the reference's
`torch.randint(n, ...)` / `torch.randperm(n)` calls consume from the global CPU generator (stdroi:343-371, 433-461).

    unpack_state(blob)        torch.get_rng_state() bytes -> int32[626] = mt state words [624], left, next
    pack_state(blob, compact) the same blob with the engine advanced to `compact` (for torch.set_rng_state)
    MT                        numpy restatement of the engine and of the two draw patterns (tests; documentation of
                              what the kernels do): draw(), skip(m), randint_first(n, n_draw, k), randperm_first(n, k)

Layout of the blob (CPUGeneratorImplState, aten/src/ATen/CPUGeneratorImpl.cpp): the legacy TH generator record -- uint64
the_initial_seed, int left, int seeded, uint64 next, uint64 state[624] (one engine word each), double normal_x, normal_y,
normal_rho, int normal_is_valid -- followed by float next_float_normal_sample and bool is_valid (5056 bytes in all;
checked against the running torch by tests/test_mt19937.py).
The engine's `operator()`: if (--left == 0) next_state(); y = state[next++]; temper(y).  `random()` is one such word;
randint(n) with n < 2^28 is random() % n, one word per element in element order (cpu_serial_kernel; from 2^28 on torch
draws two words per element -- not needed here: a candidate count is a pixel count); randperm(n) for
n < 2^32 / 20 is the forward Fisher-Yates `for i < n - 1: z = random() % (n - i); swap(r[i], r[z + i])` -- its first k
outputs are final after k draws, the other n - 1 - k draws only advance the engine."""
import numpy as np
import torch

N, M = 624, 397
_OFF_LEFT, _OFF_NEXT, _OFF_STATE = 8, 16, 24          # byte offsets inside the record (see the layout above)


def unpack_state(blob):
    """torch.get_rng_state() (uint8 [5056]) -> np.int32[626]: state[624] (bit patterns), left, next."""
    b = blob.numpy().tobytes()
    left = int(np.frombuffer(b, dtype=np.int32, count=1, offset=_OFF_LEFT)[0])
    nxt = int(np.frombuffer(b, dtype=np.uint64, count=1, offset=_OFF_NEXT)[0])
    st = np.frombuffer(b, dtype=np.uint64, count=N, offset=_OFF_STATE)
    if not (1 <= left <= N and 0 <= nxt <= N and int(st.max()) < 2 ** 32):
        raise ValueError(f"unexpected mt19937 state record (left={left}, next={nxt})")
    out = np.empty(N + 2, dtype=np.uint32)
    out[:N] = st.astype(np.uint32)
    out[N], out[N + 1] = left, nxt
    return out.view(np.int32)


def pack_state(blob, compact):
    """Copy of `blob` whose engine is at `compact` (np.int32[626] as unpack_state returns / the kernels leave it)."""
    c = np.asarray(compact).view(np.uint32)
    buf = bytearray(blob.numpy().tobytes())
    np.frombuffer(buf, dtype=np.int32, count=1, offset=_OFF_LEFT)[0] = int(c[N])
    np.frombuffer(buf, dtype=np.uint64, count=1, offset=_OFF_NEXT)[0] = int(c[N + 1])
    np.frombuffer(buf, dtype=np.uint64, count=N, offset=_OFF_STATE)[:] = c[:N].astype(np.uint64)
    return torch.frombuffer(buf, dtype=torch.uint8).clone()


class MT:
    """The engine on a compact state (numpy; the arithmetic of csrc/mt19937.hip, one draw at a time)."""

    def __init__(self, compact):
        c = np.asarray(compact).view(np.uint32).copy()
        self.st, self.left, self.next = c[:N].copy(), int(c[N]), int(c[N + 1])

    def compact(self):
        return np.concatenate((self.st, np.array([self.left, self.next], dtype=np.uint32))).view(np.int32)

    def _refill(self):
        s = [int(v) for v in self.st]

        def twist(u, v):
            return (((u & 0x80000000) | (v & 0x7FFFFFFF)) >> 1) ^ (0x9908B0DF if v & 1 else 0)

        for i in range(N - M):                                   # i + M is still old
            s[i] = s[i + M] ^ twist(s[i], s[i + 1])
        for i in range(N - M, N - 1):                            # i + M - N is already new
            s[i] = s[i + M - N] ^ twist(s[i], s[i + 1])
        s[N - 1] = s[M - 1] ^ twist(s[N - 1], s[0])
        self.st = np.array(s, dtype=np.uint32)
        self.left, self.next = N, 0

    def draw(self):
        self.left -= 1
        if self.left == 0:
            self._refill()
        y = int(self.st[self.next])
        self.next += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def skip(self, m):
        for _ in range(m):
            self.left -= 1
            if self.left == 0:
                self._refill()
            self.next += 1

    def randint_first(self, n, n_draw, k):
        """(torch.randint(n, (n_draw,)) % n)[:k]"""
        vals = [self.draw() % n for _ in range(n_draw)]
        return vals[:k]

    def randperm_first(self, n, k):
        """torch.randperm(n)[:k]  (n < 2^32 / 20)"""
        moved = {}
        for i in range(min(k, max(n - 1, 0))):
            z = self.draw() % (n - i)
            a, b = moved.get(i, i), moved.get(z + i, z + i)
            moved[i], moved[z + i] = b, a
        self.skip(max(n - 1, 0) - min(k, max(n - 1, 0)))
        return [moved.get(i, i) for i in range(min(k, n))]
