"""GPU parity of the IndexSequence codec kernels against the CPU oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(2024)
    yield "empty", np.array([], np.uint64)
    yield "zero", np.array([0], np.uint64)
    yield "dup", np.array([5, 5, 5], np.uint64)
    yield "max", np.array([0xFFFFFFFFFFFFFFFF], np.uint64)
    # deltas at the 2^15k boundaries -> 1..5 units
    d = []
    for k in (15, 30, 45, 60):
        d += [(1 << k) - 1, 1 << k, (1 << k) + 1]
    yield "boundaries", np.cumsum(np.array(d, dtype=np.uint64))
    yield "test100", np.cumsum(rng.integers(0, 0x10000, 100, dtype=np.uint64))  # IndexSequenceTest.cs:81-93
    yield "small", np.cumsum(rng.integers(1, 300, 5000, dtype=np.uint64))
    yield "mixed", np.cumsum(rng.integers(0, 1 << 33, 70001, dtype=np.uint64))
    yield "big", np.cumsum(rng.integers(1, 1 << 17, 1_000_003, dtype=np.uint64))


@pytest.mark.parametrize("name,vals", list(_cases()), ids=[c[0] for c in _cases()])
def test_encode_decode_parity(oracle_mod, name, vals):
    import docodo_amd
    want = oracle_mod.seq_encode(vals)
    got = docodo_amd.IndexSequence.from_values(vals)
    assert np.array_equal(got.units, want)
    assert np.array_equal(got.values(), vals)
    assert got.count_values() == vals.size


def test_decode_malformed_matches_oracle(oracle_mod):
    """Non-canonical input: truncated tail, 7-unit groups (C# masks the shift to 6 bits)."""
    import docodo_amd
    rng = np.random.default_rng(5)
    u = rng.integers(0, 0x10000, 50000, dtype=np.uint64).astype(np.uint16)
    u[-3:] |= 0x8000  # dangling continuation units are not yielded
    s = docodo_amd.IndexSequence(u)
    assert np.array_equal(s.values(), oracle_mod.seq_decode(u))
    assert s.count_values() == oracle_mod.seq_decode(u).size


def test_roundtrip_properties_reference_tests(oracle_mod):
    """IndexSequenceTest.cs:17-79: ConvertTest + ShiftTest through the HIP codec."""
    import docodo_amd
    rng = np.random.default_rng(11)
    vals = np.cumsum(rng.integers(0, 0x10000, 100, dtype=np.uint64))
    seq = docodo_amd.IndexSequence.from_values(np.concatenate([[0], vals]).astype(np.uint64))
    base = seq.values()
    for s in (100, 0xFFFFF, 0xFFFFFFFF):
        q = docodo_amd.IndexSequence(seq.units.copy())
        q.Shift(s)
        v = q.values()
        assert v.size == base.size and np.array_equal(v - base, np.full(base.size, s, np.uint64))
