"""Worker of tests/test_dist_cpu.py: world_size-2 gloo run of docodo_amd.dist.exchange on CPU tensors.
Every rank fabricates an export whose bytes encode (source rank, owner, kind, index) and checks that
what it receives is exactly the slices addressed to it, in source order."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DOCODO_NO_TORCH"] = "0"


def fake_export(rank, world):
    from docodo_amd.dist import META_BYTES, ShardExport
    rng = np.random.default_rng(100 + rank)
    tc = rng.integers(0, 6, world).astype(np.uint64)            # ragged, zeros allowed
    kc = (tc * rng.integers(3, 9, world).astype(np.uint64))
    uc = (tc * rng.integers(1, 50, world).astype(np.uint64))

    def fill(counts, scale, kind):
        parts = []
        for owner in range(world):
            n = int(counts[owner]) * scale
            parts.append(((np.arange(n) * 7 + rank * 31 + owner * 13 + kind * 5) % 251).astype(np.uint8))
        return torch.from_numpy(np.concatenate(parts) if parts else np.empty(0, np.uint8))

    return ShardExport(tc, kc, uc, fill(tc, META_BYTES, 0), fill(kc, 2, 1), fill(uc, 2, 2), 1000 + rank)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from docodo_amd.dist import META_BYTES, exchange
    mine = fake_export(rank, world)
    exp = [fake_export(s, world) for s in range(world)]
    for overlap in (False, True):
        check_exchange(rank, world, mine, exp, overlap)
    check_allgather(rank, world)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


def check_allgather(rank, world):
    """docodo_amd.dist.allgather_export: ragged one-owner exports (one of them empty) -> every rank holds all of them, in rank order."""
    from docodo_amd.dist import META_BYTES, ShardExport, allgather_export

    def one(r):
        n = 0 if r == 1 else 3 + 5 * r                               # rank 1 owns nothing
        f = lambda scale, kind: torch.from_numpy(((np.arange(n * scale) * 11 + r * 17 + kind * 3) % 251).astype(np.uint8))
        u = np.uint64
        return ShardExport(np.array([n], u), np.array([n * 4], u), np.array([n * 9], u), f(META_BYTES, 0), f(4 * 2, 1), f(9 * 2, 2), 500 + 7 * r)

    tc, kc, uc, meta, keys, units, mx = allgather_export(one(rank))
    exp = [one(r) for r in range(world)]
    assert mx == 500 + 7 * (world - 1)
    assert tc.tolist() == [int(e.term_count[0]) for e in exp] and kc.tolist() == [int(e.key_count[0]) for e in exp]
    assert uc.tolist() == [int(e.unit_count[0]) for e in exp]
    for got, name in ((meta, "meta"), (keys, "keys"), (units, "units")):
        assert torch.equal(got, torch.cat([getattr(e, name) for e in exp])), (rank, name)


def check_exchange(rank, world, mine, exp, overlap):
    from docodo_amd.dist import META_BYTES, exchange
    if overlap:
        # the async form distributed_merge uses: meta + keys are complete on return, the units after wait_units()
        tc, kc, uc, meta, keys, units, mx, wait_units = exchange(mine, overlap=True)
        wait_units()
    else:
        tc, kc, uc, meta, keys, units, mx = exchange(mine)
    assert mx == 1000 + world - 1
    for got, kind, scale, cnt in ((meta, 0, META_BYTES, tc), (keys, 1, 2, kc), (units, 2, 2, uc)):
        want = []
        for s in range(world):
            counts = (exp[s].term_count, exp[s].key_count, exp[s].unit_count)[kind]
            assert int(cnt[s]) == int(counts[rank])
            n = int(counts[rank]) * scale
            want.append(((np.arange(n) * 7 + s * 31 + rank * 13 + kind * 5) % 251).astype(np.uint8))
        want = np.concatenate(want) if want else np.empty(0, np.uint8)
        assert np.array_equal(got.numpy(), want), (rank, kind, overlap)


if __name__ == "__main__":
    main()
