"""Worker of tests/test_gpu_merge.py::test_distributed_merge_over_rccl_world1: the real RCCL code path of
docodo_amd.dist.distributed_merge (all-gather of sizes, three async all-to-alls, overlapped dcd_merge_begin,
dcd_merge_finish) with world_size 1 on the one GPU of the box -- same collectives, streams and waits as N > 1."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ.get("DCD_PORT", "29671"), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    import docodo_amd
    from docodo_amd import dist as ddist, synth
    n_docs = 20000
    text = synth.corpus_host(n_docs)
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    b = docodo_amd.Builder()
    b.AddPages(text, off, docs_per_page=True)
    b.Finish()
    want = b.index_image()
    for _ in range(3):                      # repeated: buffers are recycled by both allocators between rounds
        owned = ddist.distributed_merge(b, device_index=0)
        assert owned.index_image() == want
        got = ddist.gather_index(owned, dst=0, device_index=0)
        assert got.index_image() == want
    dist.barrier()
    dist.destroy_process_group()
    print("rccl world1 ok")


if __name__ == "__main__":
    main()
