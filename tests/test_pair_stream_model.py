"""The Python model of k_pair_stream (tests/helpers/pair_stream_model.py) against the oracle's '*' / '+' (IndexSequence.cs:218-322):
pins the LOGIC of the 32-bit streaming kernel on the CPU -- saturating tile-relative keys, lane summaries + carry-chain fills,
ownership, optimistic write and roll-back -- for random operands, R combinations and partition sizes."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import pair_stream_model as psm  # noqa: E402


def _seq(rng, n, g, s=0):
    if n == 0:
        return np.array([], np.uint64)
    return (np.cumsum(rng.integers(1, g, n, dtype=np.uint64)) + np.uint64(s)).astype(np.uint64)


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        na = int(rng.choice([0, 1, 3, 50, 300, 700]) * rng.uniform(0.5, 1.5))
        nb = int(rng.choice([0, 1, 3, 50, 300, 700]) * rng.uniform(0.5, 1.5))
        ga, gb = int(rng.choice([2, 5, 40, 300, 3000, 1 << 33])), int(rng.choice([2, 5, 40, 300, 3000, 1 << 33]))
        a, b = _seq(rng, na, ga, int(rng.integers(0, 1000))), _seq(rng, nb, gb, int(rng.integers(0, 1000)))
        if na and nb and rng.random() < 0.5:
            k = max(1, min(na, nb) // int(rng.choice([2, 10])))
            b = np.unique(np.concatenate([b, a[rng.integers(0, na, k)]])).astype(np.uint64)
        Ra, Rb = [(258, 259), (0, 0), (-7, -9), (-7, 258), (258, -9), (40, 3), (-1, -1), (0, -5), (100000, 1)][int(rng.integers(0, 9))]
        part = int(rng.choice([64, 300, 1024, 8192]))
        yield a, b, Ra, Rb, part


def test_fills():
    rng = np.random.default_rng(5)
    for _ in range(300):
        g, p, cin = int(rng.integers(0, 1 << 63)) | (int(rng.integers(0, 2)) << 63), int(rng.integers(0, 1 << 63)) | (int(rng.integers(0, 2)) << 63), int(rng.integers(0, 2))
        f, prev = 0, cin
        for i in range(64):
            prev = ((g >> i) & 1) | (((p >> i) & 1) & prev)
            f |= prev << i
        assert psm.fill_fwd(g, p, cin) == f
        b, nxt = 0, 0
        for i in range(63, -1, -1):
            nxt = ((g >> i) & 1) | (((p >> i) & 1) & nxt)
            b |= nxt << i
        assert psm.fill_bwd(g, p) == b


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_model_equals_oracle(oracle_mod, seed):
    o = oracle_mod
    for a, b, Ra, Rb, part in _cases(seed, 60):
        ua, ub = o.seq_encode(a), o.seq_encode(b)
        want, _ = o.seq_and(ua, Ra, ub, Rb)
        got = psm.evaluate(a, b, Ra, Rb, False, part)
        assert np.array_equal(o.seq_encode(np.array(got, np.uint64)), want), ("and", a.size, b.size, Ra, Rb, part)
        want, _ = o.seq_or(ua, Ra, ub, Rb)
        got = psm.evaluate(a, b, Ra, Rb, True, part)
        assert np.array_equal(o.seq_encode(np.array(got, np.uint64)), want), ("or", a.size, b.size, Ra, Rb, part)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_mask_form_equals_oracle(oracle_mod, seed):
    """The mask form of the kernel (lane masks per element slot, epoch-relative 32-bit ring keys) against the oracle."""
    o = oracle_mod
    stats = {}
    for a, b, Ra, Rb, part in _cases(seed, 60):
        ua, ub = o.seq_encode(a), o.seq_encode(b)
        want, _ = o.seq_and(ua, Ra, ub, Rb)
        got = psm.evaluate_masks(a, b, Ra, Rb, False, part, stats)
        assert np.array_equal(o.seq_encode(np.array(got, np.uint64)), want), ("and", a.size, b.size, Ra, Rb, part)
        want, _ = o.seq_or(ua, Ra, ub, Rb)
        got = psm.evaluate_masks(a, b, Ra, Rb, True, part, stats)
        assert np.array_equal(o.seq_encode(np.array(got, np.uint64)), want), ("or", a.size, b.size, Ra, Rb, part)
    assert stats.get("rebase", 0) > 0, "no case left its epoch: the re-staging path was not exercised"
