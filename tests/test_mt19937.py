"""attentionshift_amd/mt19937.py against torch's own CPU generator: the state record round-trips, and the engine restated
there (= the arithmetic of csrc/mt19937.hip) reproduces torch.randint / torch.randperm draw for draw, across refills,
and leaves the generator where torch leaves it."""
import numpy as np
import torch

from attentionshift_amd import mt19937 as MT


def test_state_record_round_trip_and_engine_position():
    torch.manual_seed(123)
    torch.rand(1000)                                        # somewhere inside a block
    blob = torch.get_rng_state()
    c = MT.unpack_state(blob)
    assert c.shape == (626,) and 1 <= int(c[624]) <= 624
    assert torch.equal(MT.pack_state(blob, c), blob)
    eng = MT.MT(c)
    want = torch.randint(2 ** 27, (1500,)).tolist()         # crosses at least two refills
    got = [eng.draw() % (2 ** 27) for _ in range(1500)]
    assert got == want
    # handing the advanced engine back to torch continues the same stream
    after = torch.get_rng_state()
    assert torch.equal(MT.pack_state(blob, eng.compact()), after)


def test_fresh_seed_state_has_left_one():
    torch.manual_seed(7)
    c = MT.unpack_state(torch.get_rng_state())
    eng = MT.MT(c)
    assert eng.draw() % 1000 == int(torch.randint(1000, (1,)))


def test_randint_and_randperm_patterns_match_torch_in_sequence():
    """The draw patterns of the reference-RNG mode in the order seed_pseudo_gt makes them: (2G+1) x randint(n, (n_draw,))
    then G x randperm(n)[:10], twice (two images), with counts from tens to tens of thousands."""
    torch.manual_seed(99)
    start = torch.get_rng_state()
    rng = np.random.default_rng(0)
    plan = []
    for img in range(2):
        for _ in range(7):
            n = int(rng.integers(20, 30000))
            plan.append(("randint", n, len(range(0, n, n // 20))))
        for _ in range(3):
            plan.append(("randperm", int(rng.choice([10, 11, 25, 700, 5000, 40000]))))
    want = []
    for p in plan:
        if p[0] == "randint":
            want.append((torch.randint(p[1], (p[2],)) % p[1])[:20].tolist())
        else:
            want.append(torch.randperm(p[1])[:10].tolist())
    end = torch.get_rng_state()
    eng = MT.MT(MT.unpack_state(start))
    got = [eng.randint_first(p[1], p[2], 20) if p[0] == "randint" else eng.randperm_first(p[1], 10) for p in plan]
    assert got == want
    assert torch.equal(MT.pack_state(start, eng.compact()), end)


def test_randperm_edge_sizes():
    for n in (0, 1, 2, 9, 10):
        torch.manual_seed(5 + n)
        start = torch.get_rng_state()
        want = torch.randperm(n)[:10].tolist()
        end = torch.get_rng_state()
        eng = MT.MT(MT.unpack_state(start))
        assert eng.randperm_first(n, 10) == want, n
        assert torch.equal(MT.pack_state(start, eng.compact()), end), n


def test_randint_word_width_switches_at_2_pow_28():
    """torch draws ONE engine word per element below a range of 2^28 and two (high word first) from there on; the device
    path only implements the one-word form and flags anything larger (a candidate count cannot reach 2^28 pixels)."""
    torch.manual_seed(3)
    start = torch.get_rng_state()
    for n, words in ((2 ** 28 - 1, 1), (2 ** 28, 2)):
        torch.set_rng_state(start)
        want = torch.randint(n, (5,)).tolist()
        eng = MT.MT(MT.unpack_state(start))
        d = [eng.draw() for _ in range(5 * words)]
        got = [d[i] % n for i in range(5)] if words == 1 else [((d[2 * i] << 32) | d[2 * i + 1]) % n for i in range(5)]
        assert got == want, n
