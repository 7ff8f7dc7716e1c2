"""Worker of tests/test_gpu_merge.py::test_three_ranks_on_one_gpu_give_the_one_shot_index: the whole N > 1 path -- doc-range
shards with global coordinates built on the GPU, export, all-to-all, overlapped merge, gather of the hash partitions -- with
three ranks sharing the one GPU of the box, THROUGH THE C ENTRY POINTS ONLY (round 4: dcd_comm_create,
dcd_build_merge_distributed, dcd_index_gather, dcd_index_allgather).  RCCL cannot place two ranks on one device, so the
communicator uses the library's host transport (DCD_COMM_HOST: the buffers staged through shared memory); everything else --
heavy-term balance, export, all-to-all-v, overlapped merge -- is the code that runs at N > 1 on a multi-GPU node.  torch.distributed
(gloo) only carries the communicator id once and compares digests."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import docodo_amd
    from docodo_amd import dist as ddist, synth
    per = 7000 + 500 * rank                                   # ragged shards
    first = sum(7000 + 500 * r for r in range(rank))
    total = sum(7000 + 500 * r for r in range(world))
    d = torch.empty(per * synth.DOC_BYTES, dtype=torch.uint8, device="cuda")
    synth.corpus_dev(d.data_ptr(), per, first_doc=first)
    comm = ddist.Comm.from_torch(0)                           # gloo group -> host transport; rank 0's id by one broadcast
    assert comm.transport == docodo_amd._lib.DCD_COMM_HOST and comm.n == world
    lens = comm.allgather(np.array([per * synth.DOC_BYTES], np.uint64))[:, 0]       # dcd_comm_allgather: the shards' text lengths ...
    assert int(lens[:rank].sum()) == first * synth.DOC_BYTES                       # ... whose prefix sums are the coord_base values
    for rnd in range(2):                                      # twice: both allocators recycle their buffers
        b = docodo_amd.Builder(coord_base=first * synth.DOC_BYTES)
        b.AddUniformPagesDev(d.data_ptr(), per, synth.DOC_BYTES, first_doc=first)
        b.Finish()
        owned = ddist.distributed_merge(b, comm)
        got = ddist.gather_index(owned, comm, dst=0)
        assert (got is not None) == (rank == 0)
        if rank == 0:
            text = synth.corpus_host(total)
            off = np.arange(total + 1, dtype=np.uint64) * synth.DOC_BYTES
            one = docodo_amd.Builder()
            one.AddPages(text, off, docs_per_page=True)
            one.Finish()
            assert got.index_image() == one.index_image(), "gathered .index differs from the one-shot build (round %d)" % rnd
            ddist.set_uniform_pages(got, total, synth.DOC_BYTES)
            assert got.list_image() == one.list_image()
            assert got.Count == one.stats.n_terms
            import hashlib
            want_digest = hashlib.sha256(one.index_image()).digest() + hashlib.sha256(one.list_image()).digest()
            one.close()
            got.close()
        # ... and the replica route: EVERY rank ends with the complete index (query batches shard by query)
        rep = ddist.allgather_index(owned, comm)
        ddist.set_uniform_pages(rep, total, synth.DOC_BYTES)
        import hashlib
        digest = hashlib.sha256(rep.index_image()).digest() + hashlib.sha256(rep.list_image()).digest()
        all_d = [None] * world
        dist.all_gather_object(all_d, digest)
        assert all(x == all_d[0] for x in all_d), "replicas differ between ranks"
        if rank == 0:
            assert digest == want_digest, "replica .index / .index.list differ from the one-shot build (round %d)" % rnd
        w = synth.term(3 + rank)
        res = rep.query_batch([docodo_amd.DCD_OP_TERM, rep.find(w), 255 + len(w)], [0, 3])
        assert res.status[0] == 0 and len(res.pages(0)) > 0
        rep.close()
        owned.close()
        b.close()
        dist.barrier()
    comm.close()
    dist.destroy_process_group()
    print("rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    main()
