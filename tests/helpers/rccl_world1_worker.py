"""Worker of tests/test_gpu_merge.py::test_distributed_merge_over_rccl_world1: the RCCL transport of the library's own exchange
(dcd_comm_create with an ncclUniqueId -> ncclCommInitRank; dcd_build_merge_distributed: size all-gather over ncclAllGather, the
three all-to-all-v, overlapped dcd_merge_begin, dcd_merge_finish; dcd_index_gather; dcd_index_allgather) with world_size 1 on the
one GPU of the box -- same code, streams and waits as N > 1.  No torch.distributed anywhere: what a C# host would do."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import docodo_amd
    from docodo_amd import _lib, dist as ddist, synth
    n_docs = 20000
    text = synth.corpus_host(n_docs)
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    b = docodo_amd.Builder()
    b.AddPages(text, off, docs_per_page=True)
    b.Finish()
    want = b.index_image()
    comm = ddist.Comm(1, 0, ddist.Comm.unique_id(_lib.DCD_COMM_RCCL), 0, _lib.DCD_COMM_RCCL)     # a real ncclComm of one rank
    assert comm.allgather(np.array([7, 9], np.uint64)).tolist() == [[7, 9]]
    for _ in range(3):                      # repeated: buffers are recycled between rounds
        owned = ddist.distributed_merge(b, comm)
        assert owned.index_image() == want
        got = ddist.gather_index(owned, comm, dst=0)
        assert got.index_image() == want
        rep = ddist.allgather_index(owned, comm)
        assert rep.index_image() == want
        for x in (owned, got, rep):
            x.close()
    comm.close()
    print("rccl world1 ok")


if __name__ == "__main__":
    main()
