#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Docodo hot path.

Workload (BASELINE.json configs[1]): 1 M synthetic 1 KiB documents, Zipf-1.1 over a 100 k-term
vocabulary, per GPU.  One "step" = one full index build of the resident text
(tokenise -> exact vocabulary -> stable radix sort by term -> 15-bit var-ushort pack -> ".index"
image in HBM), followed -- outside the step timing -- by the single-term query batch of the config.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` = text bytes indexed by all ranks / wall time of the K
timed steps (barrier + synchronize on both sides, max over ranks), text already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0


def algorithmic_bytes(kernel, T, N, U, n_launch=1, n_hot=0, id_keys=False):
    """Algorithmic HBM bytes of ONE launch, averaged over the kernel's launches of a step.
    With the hot / cold split the radix kernels run once over all N tuples and (n_launch - 1) times over the N - n_hot
    cold ones; an id-keyed build's k_img_units also writes the units in term order (one read, two writes)."""
    n_launch = max(1, int(round(n_launch)))
    tuples = (N + (n_launch - 1) * (N - n_hot)) / n_launch if n_hot else N
    return {
        "k_tok_count": T,                 # reads the text once
        "k_tok_emit": T + 8 * N,          # reads the text, writes one (term, coord) tuple per posting
        "k_rs_hist": 4 * tuples,          # reads the keys
        "k_rs_scatter": 16 * tuples,      # reads and writes every 8-byte tuple
        "k_enc_reduce": 8 * N,            # reads the sorted tuples
        "k_enc_write": 8 * N + 2 * U,     # reads the tuples, writes the ushort units
        "k_img_units": (6 if id_keys else 4) * U,   # copies the units into the .index image (+ into term order)
    }.get(kernel, 0)


# what actually limits each kernel (PMC runs): the roofline above prices every kernel against HBM, as the
# contract asks, but the tokeniser kernels are bound by VALU issue, not by bytes
LIMITER = {"k_tok_emit": "VALU issue + the latency of one table gather per token (profiles/r02_pmc_build.txt; scripts/diag_tok.py: front half -- "
                         "classification, boundaries, token list -- 0.53 ms, hashing + tuple stores +0.30 ms, table lookup +0.35 ms)",
           "k_tok_count": "VALU issue (PMC: ~100 % busy; 280 wave-instructions per wave tile)",
           "k_rs_scatter": "VALU issue (match-any ballot ranking, ~50 % busy) + hbm (first pass: partial lines of short digit runs)",
           "k_rs_hist": "LDS atomics + hbm"}


TRAFFIC_FILE = "r02_final_traffic.json"


def measured_traffic(kernel, T, N, U):
    """HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, stored in
    profiles/r02_final_traffic.json by scripts/traffic_json.py).  FETCH_SIZE counts a streamed read at half its bytes
    on gfx950 (MI355X_MICROARCH.md, HBM): the streamed part of the kernel's reads (its algorithmic read bytes) is doubled,
    what it fetches beyond that (table gathers) is taken as counted.  Only valid for the configuration it was measured
    on; None otherwise."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if not os.path.exists(path):
        return None
    tr = json.load(open(path))
    if (tr.get("T"), tr.get("N"), tr.get("U")) != (T, N, U):
        return None
    # names as rocprofv3 prints them (per-launch maxima, i.e. the full-size tuple passes of the radix kernels)
    names = {"k_rs_scatter": ("k_rs_scatter<",), "k_rs_hist": ("k_rs_hist",), "k_enc_write": ("k_pack_write",),
             "k_enc_reduce": ("k_pack_count",), "k_tok_emit": ("k_tok_emit<",), "k_tok_count": ("k_tok_count<",),
             "k_img_units": ("k_img_units",)}
    pref = names.get(kernel, (kernel,))
    hits = [v["traffic_bytes"] for n, v in tr["kernels"].items() if n.startswith(pref)]
    if not hits:
        return None
    # per launch, like `achieved`: the radix kernels run under several instantiations per step (pass A over all tuples, the
    # cold passes over the rest; the file keeps the largest launch of every instantiation) -- their mean; the tiny launches
    # of the vocabulary's string sort (other digit widths, < 5 % of the largest) do not count
    big = [h for h in hits if h >= 0.05 * max(hits)]
    return int(sum(big) / len(big))


def cpu_baseline(n_docs_sample):
    """Reference-faithful CPU restatement (oracle/, single thread) on a bounded sample of the same
    corpus: text bytes / (index + serialise) seconds.  Baseline, not the target."""
    import oracle
    from docodo_amd import synth
    text = synth.corpus_host(n_docs_sample)
    off = np.arange(n_docs_sample + 1, dtype=np.uint64) * synth.DOC_BYTES
    ix = oracle.Index(stemmers=[])
    b = oracle.Builder(ix)
    t0 = time.perf_counter()
    b.index_docs_latin1(text, off)
    img, lst = b.save()
    dt = time.perf_counter() - t0
    return {"value": round(text.size / 1e9 / dt, 5), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "first %d docs (%.1f MB) of the same synthetic corpus, oracle Builder single thread, %.1f s"
                      % (n_docs_sample, text.size / 1e6, dt)}


def inflight_bench(n_threads, steps, text, D, device, T):
    """EXTRA figure, never `value`: the reference runs MaxDegreeOfParallelism (default 2) index tasks at once, each with a
    Builder of its own (Index.cs:127, 464-465, Build.cs:452).  Here: n_threads host threads, each building the resident
    text `steps` times through its own handle (own HIP streams); the VALU-bound tokeniser of one build overlaps the
    memory-bound sort / pack of the other.  Aggregate text bytes / wall time."""
    import threading
    import docodo_amd
    from docodo_amd import synth

    def worker():
        for _ in range(steps):
            b = docodo_amd.Builder(device=device)
            b.AddUniformPagesDev(text.data_ptr(), D, synth.DOC_BYTES)
            b.Finish()
            b.close()

    worker()                                    # warm-up: the second handle's buffers
    torch.cuda.synchronize()
    ts = [threading.Thread(target=worker) for _ in range(n_threads)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"builders": n_threads, "builds": n_threads * steps, "GB_per_s": round(n_threads * steps * T / 1e9 / dt, 2),
            "ms_per_build": round(dt / (n_threads * steps) * 1e3, 3),
            "note": "extra: independent build handles driven from %d host threads (the reference's MaxDegreeOfParallelism); `value` is the one-builder figure" % n_threads}


def query_bench(ix, n_single, n_and3, cpu_sample, img=None, lst=None, seed_offset=0):
    """configs[1] query leg (10 k single-term queries over ranks 101..100k, Appendix F) plus a batch of
    3-term AND queries (opt == null -> Dist = 255).  Device time of dcd_query_batch (HIP events), results
    produced on the device; the D2H copy of the result arrays is excluded.  CPU: oracle, 1 thread, bounded."""
    import docodo_amd
    from docodo_amd import synth
    T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
    out = {}
    for name, nq, nt in (("single_term", n_single, 1), ("and3", n_and3, 3)):
        ranks = synth.query_ranks(nq, nt, seed=synth.SEED_QUERY + seed_offset)
        prog, off, npost, words = [], [0], 0, []
        for row in ranks:
            ws = [synth.term(int(k)) for k in row]
            ts = [ix.find(w) for w in ws]
            if any(t is None for t in ts):
                continue
            for k, (w, t) in enumerate(zip(ws, ts)):
                prog += [T, t, 255 + len(w)]
                npost += ix.term_count(t)
                if k:
                    prog.append(A)
            off.append(len(prog))
            words.append(ts)
        prog, off = np.array(prog, np.int32), np.array(off, np.uint64)
        ix.query_batch(prog[:off[min(8, off.size - 1)]], off[:min(9, off.size)], keep_handle=True)   # warm-up (decodes the index once)
        ms = min(ix.query_batch(prog, off, keep_handle=True) for _ in range(3))
        nq_eff = off.size - 1
        out[name] = {"queries": nq_eff, "postings_in": int(npost), "device_ms": round(ms, 3),
                     "queries_per_s": round(nq_eff / (ms / 1e3), 0), "gpostings_per_s": round(npost / 1e9 / (ms / 1e3), 2)}
        if cpu_sample and img is not None:
            import oracle
            oix = oracle.Index(stemmers=[])
            oix.load(img, lst)
            k = min(cpu_sample, nq_eff)
            t0 = time.perf_counter()
            oix.query_batch(prog[:off[k]], off[:k + 1])
            dt = time.perf_counter() - t0
            out[name]["cpu_baseline_queries_per_s"] = round(k / dt, 1)
            out[name]["cpu_baseline_sample"] = "first %d queries, oracle 1 thread, %.1f s" % (k, dt)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=1_000_000, help="documents per GPU (1 KiB each)")
    ap.add_argument("--cpu-docs", type=int, default=500_000, help="documents in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--queries", type=int, default=10_000, help="single-term queries of configs[1] (0 = skip the query leg)")
    ap.add_argument("--and3", type=int, default=20_000, help="3-term AND queries in the extra query batch")
    ap.add_argument("--inflight", type=int, default=2, help="extra (untimed for `value`) leg: this many builders at once, as the "
                    "reference's MaxDegreeOfParallelism index tasks (Index.cs:127, 464-465); 0/1 = skip")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("DOCODO_BENCH_PIPELINE", "1")),
                    help="N > 1: 1 = the exchange + merge of step i runs on a second host thread beside the build of step i + 1 "
                    "(every merge completes inside the timed region); 0 = one after the other")
    ap.add_argument("--lean-events", type=int, default=0, help="1: the timed steps record HIP events around the dominant kernel only "
                    "(chosen in the fully instrumented warm-up steps: +1-2 %% value); 0 (default): every event in the timed steps")
    ap.add_argument("--force-exchange", action="store_true", help="rehearsal: run the N > 1 step (exchange + merge) with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    backend = "nccl"                                   # = RCCL on ROCm
    if os.environ.get("DOCODO_BENCH_REHEARSAL"):       # N ranks on ONE GPU over gloo: plumbing rehearsal, not a measurement
        local_rank, backend = 0, "gloo"
    torch.cuda.set_device(local_rank)
    dist = None
    exchange = world > 1 or args.force_exchange
    if exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29671")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import docodo_amd
    from docodo_amd import synth
    from docodo_amd._lib import KERNEL_NAMES

    D = args.docs
    T = D * synth.DOC_BYTES
    text = torch.empty(T, dtype=torch.uint8, device="cuda")
    first_doc = rank * D                               # doc-range shard of this rank (weak scaling)
    synth.corpus_dev(text.data_ptr(), D, first_doc=first_doc, device=local_rank)
    coord_base = first_doc * synth.DOC_BYTES           # shards encode GLOBAL coordinates

    from docodo_amd import dist as ddist
    merge_ms = []

    def merge(b):
        # exchange step: per-term segments travel to their owner (hash(key) % world) with RCCL
        # all-to-all over xGMI and are merged there; part of the timed step
        t0 = time.perf_counter()
        owned = ddist.distributed_merge(b, device_index=local_rank)
        merge_ms.append((time.perf_counter() - t0) * 1e3)      # (host time of the calls; the device side is covered by the fence)
        owned.close()
        b.close()

    # N > 1, pipelined: ONE worker thread takes the finished builders in step order and runs their exchange + merge while the
    # main thread builds the next shard.  Every collective inside the timed region is issued by that thread, in the same
    # order on every rank; the main thread joins it before it meets the other ranks at a barrier.
    worker = None
    if exchange and args.pipeline:
        import queue
        import threading
        jobs = queue.Queue(maxsize=1)          # (at most one merge waiting: bounds the memory of finished builders)
        failed = []

        def work():
            torch.cuda.set_device(local_rank)
            while True:
                b = jobs.get()
                try:
                    if b is None:
                        return
                    if not failed:
                        merge(b)
                except BaseException as e:     # (keep consuming: the main thread must never block on a full queue)
                    failed.append(e)
                finally:
                    jobs.task_done()

        worker = threading.Thread(target=work, daemon=True)
        worker.start()

    def drain():
        if worker is not None:
            jobs.join()
            if failed:
                raise failed[0]

    def step():
        # (N > 1: the shard is exported and merged, never serialised on its own -- no ".index" image, no term-ordered copy)
        b = docodo_amd.Builder(coord_base=coord_base, device=local_rank, shard_only=exchange)
        b.AddUniformPagesDev(text.data_ptr(), D, synth.DOC_BYTES, first_doc=first_doc)
        st = b.Finish()
        if not exchange:
            b.close()
        elif worker is not None:
            jobs.put(b)
        else:
            merge(b)
            torch.cuda.synchronize()
        return st

    def fence():
        drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Events.  Every HIP event a build records is a marker packet its stream stops at; the ~30 of a fully instrumented build
    # (stage timers + probes around every kernel) cost 0.12 ms of a 4 ms step.  The warm-up steps run fully instrumented and
    # give the per-kernel and per-stage times of the report and the DOMINANT kernel; the timed steps then record events
    # around that kernel only (dcd_set_build_events(0, 1 << k)): its launch durations -- roofline.achieved -- are measured
    # live in the timed region, everything else comes from the warm-up steps.  That is --lean-events 1; the default keeps every
    # event in the timed region (measured: 4.02-4.07 ms per step against 3.99-4.00).
    cal_k, cal_n, cal_stage, cal_steps = np.zeros(8), np.zeros(8), np.zeros(6), 0
    for i in range(args.warmup):
        st = step()
        if i > 0 or args.warmup == 1:          # (the first step of a process also allocates)
            cal_k += np.array(list(st.ms_kernel))
            cal_n += np.array(list(st.n_launch))
            cal_stage += np.array([st.ms_tokenize, st.ms_vocab, st.ms_sort, st.ms_pack, st.ms_image, st.ms_total])
            cal_steps += 1
    lean = bool(args.lean_events) and cal_steps > 0
    if lean:
        docodo_amd.set_build_events(0, 1 << int(np.argmax(cal_k[:7])))
    fence()
    t0 = time.perf_counter()
    ksum = np.zeros(8)
    kcnt = np.zeros(8)
    stage = np.zeros(6)
    last = None
    for _ in range(args.steps):
        st = step()
        ksum += np.array(list(st.ms_kernel))
        kcnt += np.array(list(st.n_launch))
        stage += np.array([st.ms_tokenize, st.ms_vocab, st.ms_sort, st.ms_pack, st.ms_image, st.ms_total])
        last = (st.n_postings, st.n_terms, st.n_units, st.index_bytes, st.sort_passes, st.id_keys, st.n_hot_terms, st.n_hot_postings)
    fence()
    dt = time.perf_counter() - t0
    if lean:
        docodo_amd.set_build_events(2, 0)
    if worker is not None:
        jobs.put(None)                     # (the worker leaves)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        N, V, U, img_bytes, passes, id_keys, hot_terms, hot_post = last
        dom_timed = (ksum.copy(), kcnt.copy())
        if lean:                           # the report's per-kernel / per-stage figures: the fully instrumented warm-up steps
            ksum, kcnt, stage = cal_k * args.steps / cal_steps, cal_n * args.steps / cal_steps, cal_stage * args.steps / cal_steps
        lps = kcnt / args.steps            # launches per step of every kernel

        def abytes_of(k):
            return algorithmic_bytes(KERNEL_NAMES[k], T, N, U, lps[k], hot_post, bool(id_keys))
        ms_step = dt / args.steps * 1e3
        value = world * T / 1e9 / (dt / args.steps)
        # dominant kernel = largest total device time per step (HIP events on the build stream)
        per_step = ksum / args.steps
        dom = int(np.argmax(per_step[:7]))
        name = KERNEL_NAMES[dom]
        avg_ms = dom_timed[0][dom] / max(1.0, dom_timed[1][dom])      # (the timed region's own launches of the dominant kernel)
        abytes = abytes_of(dom)
        achieved = abytes / 1e9 / (avg_ms / 1e3)
        out = {
            "metric": "text_indexed_GB_per_s", "value": round(value, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic 1 KiB docs per GPU, Zipf-1.1 over 100k terms (SURVEY App. F); "
                                   "full index build, text resident in HBM -> .index image resident in HBM" % D,
                       "docs_per_gpu": D, "text_bytes_per_gpu": T, "postings": int(N), "terms": int(V), "units": int(U),
                       "index_bytes": int(img_bytes), "sort_passes": int(passes), "sort": (
                           ("by the term ids the vocabulary table hands out while the tokeniser fills it (tuples carry their sort keys); "
                            "ordinal vocabulary sort on a second stream" + ("; %d hot terms chosen by a sampling launch (%.1f %% of the postings) "
                            "final after the first radix pass" % (hot_terms, 100.0 * hot_post / max(1, N)) if hot_terms else "")) if id_keys == 2 else
                           "by dense vocabulary id, ordinal vocabulary sort on a second stream; %d hot terms (%.1f %% of the postings) final "
                           "after the first radix pass" % (hot_terms, 100.0 * hot_post / max(1, N))) if id_keys else "by ordinal rank",
                       "hot_postings": int(hot_post),
                       "sharding": "doc-range, global coordinates"},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_copy": round(achieved / HBM_COPY_GBS, 4),
                         "algorithmic_bytes_per_launch": int(abytes), "avg_launch_ms": round(avg_ms, 4),
                         "launches_per_step": round(kcnt[dom] / args.steps, 2), "traffic": measured_traffic(name, T, int(N), int(U)),
                         "limiter": LIMITER.get(name, "hbm")},
            # (k_tok_count does not run in plain text builds: the tokeniser writes tile-strided tuples without a counting pass)
            "roofline_all": {KERNEL_NAMES[k]: round(abytes_of(k) / 1e9 / (ksum[k] / kcnt[k] / 1e3) / HBM_PEAK_GBS, 4)
                             for k in range(7) if kcnt[k] > 0},
            "device_ms_per_step": {"tokenize": round(stage[0] / args.steps, 3), "vocab": round(stage[1] / args.steps, 3),
                                   "sort": round(stage[2] / args.steps, 3), "pack": round(stage[3] / args.steps, 3),
                                   "image": round(stage[4] / args.steps, 3), "total": round(stage[5] / args.steps, 3)},
            "kernel_ms_per_step": {KERNEL_NAMES[k]: round(per_step[k], 4) for k in range(8)},
            "events": ("timed steps: HIP events around the dominant kernel only (roofline.avg_launch_ms); roofline_all, device_ms_per_step and "
                       "kernel_ms_per_step: the %d fully instrumented warm-up step(s)" % cal_steps) if lean else "every event in the timed steps",
            "postings_per_s": round(world * N / (dt / args.steps), 0),
        }
        if exchange:
            out["exchange_merge_ms_per_step_rank0"] = round(float(np.mean(merge_ms[args.warmup:])), 3)
            out["config"]["exchange"] = ("RCCL all-to-all of per-term segments (meta, keys, units) to owner = hash(key) % world; dcd_merge_begin overlaps "
                                         "the unit exchange, then dcd_merge_finish" +
                                         ("; the exchange + merge of step i runs on a second host thread beside the build of step i + 1, all of them inside the timed region"
                                          if worker is not None else ""))
        if world == 1 and args.inflight > 1 and not exchange:
            out["builders_in_flight"] = inflight_bench(args.inflight, args.steps, text, D, local_rank, T)
        if world == 1 and args.cpu_docs > 0 and not exchange:
            out["cpu_baseline"] = cpu_baseline(min(args.cpu_docs, D))
        if world == 1 and args.queries > 0 and not exchange:
            # query leg of configs[1], outside the timed build steps
            bq = docodo_amd.Builder(coord_base=coord_base, device=local_rank)
            bq.AddUniformPagesDev(text.data_ptr(), D, synth.DOC_BYTES, first_doc=first_doc)
            bq.Finish()
            img, lst = (bq.index_image(), bq.list_image()) if args.cpu_docs > 0 else (None, None)
            ixq = bq.Build()
            out["queries"] = query_bench(ixq, args.queries, args.and3, 3000 if args.cpu_docs > 0 else 0, img, lst)
            ixq.close()
            bq.close()
    if world > 1 and args.queries > 0:
        # query-sharded leg: every rank holds a replica of the configs[1] index (docs 0..D) and evaluates
        # its own batches (different query seeds per rank); no exchange.  Aggregate = all queries / slowest rank.
        synth.corpus_dev(text.data_ptr(), D, first_doc=0, device=local_rank)
        bq = docodo_amd.Builder(device=local_rank)
        bq.AddUniformPagesDev(text.data_ptr(), D, synth.DOC_BYTES)
        bq.Finish()
        ixq = bq.Build()
        qr = query_bench(ixq, args.queries, args.and3, 0, seed_offset=rank)
        ixq.close()
        bq.close()
        agg = {}
        for name in ("single_term", "and3"):
            cdev = "cuda" if backend == "nccl" else "cpu"
            v = torch.tensor([qr[name]["queries"], qr[name]["postings_in"]], dtype=torch.float64, device=cdev)
            m = torch.tensor([qr[name]["device_ms"]], dtype=torch.float64, device=cdev)
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            ms = float(m.item())
            agg[name] = {"queries": int(v[0].item()), "postings_in": int(v[1].item()), "device_ms_slowest_rank": round(ms, 3),
                         "queries_per_s": round(float(v[0].item()) / (ms / 1e3), 0),
                         "gpostings_per_s": round(float(v[1].item()) / 1e9 / (ms / 1e3), 2),
                         "sharding": "by query, index replicated on every GPU"}
        if rank == 0:
            out["queries"] = agg
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
