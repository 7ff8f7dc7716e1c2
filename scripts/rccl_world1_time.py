"""Times docodo_amd.dist.distributed_merge (export -> RCCL all-to-all -> overlapped merge) with world_size 1 on one GPU at
the bench's shard size: the host + collective-launch + merge cost of the N > 1 step without the fabric transfer."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29672", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import docodo_amd
    from docodo_amd import dist as ddist, synth
    text = torch.empty(docs * synth.DOC_BYTES, dtype=torch.uint8, device="cuda")
    synth.corpus_dev(text.data_ptr(), docs, first_doc=0, device=0)
    b = docodo_amd.Builder(device=0)
    b.AddUniformPagesDev(text.data_ptr(), docs, synth.DOC_BYTES, first_doc=0)
    b.Finish()
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = ddist.export_shard(b, 1)
        t1 = time.perf_counter()
        tc, kc, uc, meta, keys, units, mx, wait_units = ddist.exchange(e, None, overlap=True)
        t2 = time.perf_counter()
        pend = ddist.PendingMerge(tc, kc, uc, meta, keys, mx, 0)
        t3 = time.perf_counter()
        wait_units()
        t4 = time.perf_counter()
        ix = pend.finish(units)
        t5 = time.perf_counter()
        print("rep %d: export %.2f  exchange(meta+keys landed) %.2f  merge_begin %.2f  wait_units %.2f  merge_finish %.2f  total %.2f ms"
              % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3), flush=True)
        ix.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
