"""Worker of tests/test_dist_cpu.py: N processes rendezvous on a dcd_comm with the HOST transport (the 128-byte id travels over a
gloo broadcast, as it would over any channel a host has) and run the library's small-blob all-gather -- the collective that carries
the split sizes, MaxCoord and the heavy-term lists of dcd_build_merge_distributed -- with ragged timing and payload sizes.  No GPU:
the communicators are host-only (device -1); the device-buffer collectives are covered by the GPU tests."""
import os
import sys
import time

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from docodo_amd import _lib
    from docodo_amd.dist import Comm
    comm = Comm.from_torch(-1)
    assert comm.transport == _lib.DCD_COMM_HOST and (comm.n, comm.rank) == (world, rank)
    rng = np.random.default_rng(7)
    for it in range(40):
        n = int(rng.integers(1, 5000))                       # (the same size on every rank: one generator, one seed)
        if it % 7 == rank:
            time.sleep(0.01 * rank)                          # ragged arrival
        mine = (np.arange(n, dtype=np.uint64) * 3 + 1000 * rank + it).astype(np.uint64)
        got = comm.allgather(mine)
        assert got.shape == (world, n)
        for r in range(world):
            assert np.array_equal(got[r], np.arange(n, dtype=np.uint64) * 3 + 1000 * r + it), (it, r)
    # the coord_base values of a doc-range sharded build: prefix sums of the all-gathered text lengths
    lens = comm.allgather(np.array([1024 * (100 + rank)], np.uint64))[:, 0]
    assert int(lens[:rank].sum()) == sum(1024 * (100 + r) for r in range(rank))
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
