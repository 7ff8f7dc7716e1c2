#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Docodo hot path.

Metric (BASELINE.json): GB/s of text indexed + 3-term AND queries/s, with the fraction of the HBM roofline.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (a CHILD
torch.distributed.run, before anything here has touched HIP) and forwards rank 0's line.

Legs of one run (ONE JSON line on rank 0):
  value / ms_per_step / roofline   configs[1]: K timed full builds of 1 M synthetic 1 KiB documents per GPU, text resident in
                                   HBM -> ".index" image resident in HBM (N > 1: weak scaling, every step ends with the RCCL
                                   exchange + merge of the per-term segments, shards built shard-only -- see config.shard_only)
  general_pages                    the same build through dcd_build_add_pages_dev with EXPLICIT page offsets (what a real host
                                   calls; the headline uses the arithmetic uniform pages of the synthetic corpus)
  cpu_baseline                     the oracle (reference-faithful CPU restatement) on a bounded sample: 1 core, and all cores
                                   (doc-range shards built in parallel + the reference's pairwise MergeFiles)
  queries                          configs[1]: 10 k single-term queries (+ a 3-term AND batch) on the 1 M-doc index
  configs234                       configs[2]: the 10 M-doc index (N = 1: three doc-range segments merged on the device;
                                   N > 1: strong scaling, 10 M / N docs per rank, exchange + merge + replica all-gather);
                                   configs[3]: 100 k three-term AND queries (Dist = 255); configs[4]: 100 k exact phrases --
                                   sharded by query over the replicas; each with its HBM-roofline fraction on the SURVEY 8(d)
                                   bytes A_query = 2 sum(U_t) + 4 N_pos + 8 N_pages and an oracle sample on the same index.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

_T0 = time.perf_counter()


def note(msg):
    """progress line on stderr (a long run must show that it is alive; also the per-leg wall times of a run)"""
    if os.environ.get("RANK", "0") == "0":
        print("[bench %7.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0


def algorithmic_bytes(kernel, T, N, U, n_launch=1, n_hot=0, id_keys=False, passes=0, hot32=False):
    """Algorithmic HBM bytes of ONE launch, averaged over the kernel's launches of a step.
    With the hot / cold split the radix kernels run once over all N tuples and (n_launch - 1) times over the N - n_hot
    cold ones; an id-keyed build's k_img_units also writes the units in term order (one read, two writes).  Round 3: the
    histogram kernel no longer runs for the first pass (the tokeniser counts its digits) nor for the first cold pass (the first
    pass's scatter kernel counts them): fewer histogram launches than passes = every one of them reads cold tuples only."""
    n_launch = max(1, int(round(n_launch)))
    tuples = (N + (n_launch - 1) * (N - n_hot)) / n_launch if n_hot else N
    if kernel == "k_rs_hist" and n_hot and passes and n_launch < passes:
        tuples = N - n_hot
    # round 4 (hot32): the hot buckets leave the first pass as 4-byte coordinates -- that pass writes, and the pack kernels read,
    # 4 instead of 8 bytes for every hot posting
    h4 = 4 * n_hot if (hot32 and n_hot) else 0
    if kernel == "k_rs_scatter":
        return 16 * tuples - h4 / n_launch
    if kernel == "k_enc_reduce":
        return 8 * N - h4
    if kernel == "k_enc_write":
        return 8 * N - h4 + 2 * U
    return {
        "k_tok_count": T,                 # reads the text once
        "k_tok_emit": T + 8 * N,          # reads the text, writes one (term, coord) tuple per posting
        "k_rs_hist": 4 * tuples,          # reads the keys
        "k_rs_scatter": 16 * tuples,      # reads and writes every 8-byte tuple
        "k_enc_reduce": 8 * N,            # reads the sorted tuples
        "k_enc_write": 8 * N + 2 * U,     # reads the tuples, writes the ushort units
        "k_img_units": 4 * U,             # copies the units into the .index image (round 3: the term-ordered posting array of a
                                          # queryable index is made when a handle adopts the build, not in every build)
    }.get(kernel, 0)


# what actually limits each kernel (PMC runs): the roofline above prices every kernel against HBM, as the
# contract asks, but the tokeniser kernels are bound by VALU issue, not by bytes
LIMITER = {"k_tok_emit": "VALU issue + the latency of one table gather per token (profiles/r02_pmc_build.txt; scripts/diag_tok.py: front half -- "
                         "classification, boundaries, token list -- 0.53 ms, hashing + tuple stores +0.30 ms, table lookup +0.35 ms)",
           "k_tok_count": "VALU issue (PMC: ~100 % busy; 280 wave-instructions per wave tile)",
           "k_rs_scatter": "VALU issue (match-any ballot ranking, ~50 % busy) + hbm (first pass: partial lines of short digit runs)",
           "k_rs_hist": "LDS atomics + hbm"}
LIMITER["k_tok_emit"] = ("VALU issue + the latency of one table gather per token (profiles/r02_pmc_build.txt: SQ_ACTIVE_INST_VALU ~87 % of the busy cycles; "
                         "scripts/diag_tok.py: front half -- classification, boundaries, token list -- 0.53 ms, hashing + tuple stores +0.30 ms, table "
                         "lookup +0.35 ms; round 3 adds the first radix pass's digit counts, +0.04 ms, which removes a 0.19 ms histogram kernel)")


TRAFFIC_FILES = ("r04_final_traffic.json", "r03_final_traffic.json", "r02_final_traffic.json")


def measured_traffic(kernel, T, N, U):
    """HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, stored in
    profiles/rNN_final_traffic.json by scripts/traffic_json.py).  FETCH_SIZE counts a streamed read at half its bytes
    on gfx950 (MI355X_MICROARCH.md, HBM): the streamed part of the kernel's reads (its algorithmic read bytes) is doubled,
    what it fetches beyond that (table gathers) is taken as counted.  Only valid for the configuration it was measured
    on; None otherwise."""
    tr = None
    for f in TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", f)
        if os.path.exists(path):
            cand = json.load(open(path))
            if (cand.get("T"), cand.get("N"), cand.get("U")) == (T, N, U):
                tr = cand
                break
    if tr is None:
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


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle/): reported beside the GPU numbers, never the target
# ------------------------------------------------------------------------------------------------------------------
def _oracle_shard_image(text, first_doc):
    import oracle
    from docodo_amd import synth
    n_docs = text.size // synth.DOC_BYTES
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    ix = oracle.Index(stemmers=[])
    b = oracle.Builder(ix)
    b.index_docs_latin1(text, off, first_doc=first_doc)
    return b.save()


def cpu_baseline(n_docs_1core, n_docs_all):
    """Reference-faithful CPU restatement (oracle/) on bounded samples of the same corpus: text bytes / (index + serialise)
    seconds.  `value` = ONE thread (the canonical, deterministic reference configuration: one index task).  `all_cores`:
    the reference runs MaxDegreeOfParallelism index tasks and merges their indexes (Index.cs:127, 464-465, 488-500) -- here one
    doc-range shard per hardware thread (oracle Builder; ctypes releases the GIL) and the reference's own two-input
    MergeFiles with coordinate shifts, pairwise, the rounds in parallel as well."""
    import concurrent.futures as cf
    import oracle
    from docodo_amd import synth
    oracle.lib()
    text = synth.corpus_host(n_docs_1core)          # (corpus generation is outside the timed region on both sides)
    t0 = time.perf_counter()
    _oracle_shard_image(text, 0)
    dt1 = time.perf_counter() - t0
    del text
    out = {"value": round(n_docs_1core * synth.DOC_BYTES / 1e9 / dt1, 5), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": "first %d docs (%.1f MB) of the same synthetic corpus, oracle Builder single thread, %.1f s"
                     % (n_docs_1core, n_docs_1core * synth.DOC_BYTES / 1e6, dt1)}
    C = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    C = max(1, min(C, 64))
    if n_docs_all > 0 and C > 1:
        per = max(1, n_docs_all // C)
        n_all = per * C
        with cf.ThreadPoolExecutor(C) as ex:
            texts = list(ex.map(lambda g: synth.corpus_host(per, first_doc=g * per), range(C)))
            t0 = time.perf_counter()
            imgs = list(ex.map(lambda g: _oracle_shard_image(texts[g], g * per)[0], range(C)))
            tb = time.perf_counter()
            # every shard into the global coordinate space (IndexSequence.Shift of each list, Index.cs:671), then pairwise
            # MergeFiles rounds (a two-input merge is free of the reference's >= 3-input quirk, oracle.cpp)
            imgs = list(ex.map(lambda g: oracle.merge_files([imgs[g]], True, [g * per * synth.DOC_BYTES]), range(C)))
            while len(imgs) > 1:
                pairs = [imgs[i:i + 2] for i in range(0, len(imgs), 2)]
                imgs = list(ex.map(lambda p: oracle.merge_files(p, False) if len(p) == 2 else p[0], pairs))
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": round(n_all * synth.DOC_BYTES / 1e9 / dt, 5), "unit": "GB/s", "cores": C,
                            "sample": "first %d docs (%.1f MB): %d doc-range shards built in parallel (%.1f s) + Shift and pairwise MergeFiles "
                                      "(%.1f s), oracle" % (n_all, n_all * synth.DOC_BYTES / 1e6, C, tb - t0, dt - (tb - t0))}
    return out


# ------------------------------------------------------------------------------------------------------------------
# query legs
# ------------------------------------------------------------------------------------------------------------------
class TermInfo:
    """find / postings / units per word, cached (each costs one ctypes call)."""

    def __init__(self, ix):
        self.ix, self.c = ix, {}

    def __call__(self, w):
        r = self.c.get(w)
        if r is None:
            t = self.ix.find(w)
            r = (t, self.ix.term_count(t), self.ix.term_units(t)) if t is not None else (None, 0, 0)
            self.c[w] = r
        return r


def run_query_leg(ix, queries, info, chunk, label, reps=1, warm=True):
    """queries: list of [(word, R), ...] joined by AND.  Evaluated in calls of `chunk` queries (bounds the host-side result
    arrays); device time of dcd_query_batch (HIP events), results produced on the device, D2H of the result arrays excluded.
    Roofline on the SURVEY 8(d) bytes: 2 * units of every operand list + 4 * positions + 8 * pages, exact counts."""
    import docodo_amd
    T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
    tot_ms, nq, npost, nunits, n_pages, n_pos, per, wall_ms = 0.0, 0, 0, 0, 0, 0, [], 0.0
    ops_named, ops_eval, un_named, un_eval = 0, 0, 0, 0
    progs = []
    for c0 in range(0, len(queries), chunk):
        prog, off = [], [0]
        for q in queries[c0:c0 + chunk]:
            ts = [info(w) for w, _ in q]
            if any(t[0] is None for t in ts):
                continue
            for k, ((w, R), (t, cnt, un)) in enumerate(zip(q, ts)):
                prog += [T, t, R]
                npost += cnt
                nunits += un
                if k:
                    prog.append(A)
            off.append(len(prog))
        if len(off) == 1:
            continue
        pa, oa = np.array(prog, np.int32), np.array(off, np.uint64)
        progs.append((pa, oa))
        if not progs[1:] and warm:
            # the first call of a leg also sizes the handle's work arena (one hipMalloc of up to half the device memory: seconds):
            # run it once untimed, as a serving process would have done long before
            tw = time.perf_counter()
            ix.query_batch(pa, oa, totals=True)
            note("  %s: warm-up call (work arena): %.1f s" % (label[:12], time.perf_counter() - tw))
        best = None
        for _ in range(reps):
            tw = time.perf_counter()
            r = ix.query_batch(pa, oa, totals=True)
            r["wall_ms"] = (time.perf_counter() - tw) * 1e3
            if best is None or r["ms"] < best["ms"]:
                best = r
        wall_ms += best["wall_ms"]
        tot_ms += best["ms"]
        nq += len(off) - 1
        note("  %s: call %d: %d queries, %.1f ms device" % (label[:12], len(per) + 1, len(off) - 1, best["ms"]))
        n_pages += best["n_pages"]
        n_pos += best["n_pos"]
        ops_named += best.get("ops_named", 0)
        ops_eval += best.get("ops_evaluated", 0)
        un_named += best.get("units_named", 0)
        un_eval += best.get("units_evaluated", 0)
        per.append(round(best["ms"], 2))
    a_query = 2 * nunits + 4 * n_pos + 8 * n_pages
    a_cse = a_query - 2 * (un_named - un_eval)        # every distinct operator's index-term operands once
    out = {"queries": nq, "postings_in": int(npost), "units_in": int(nunits), "positions_out": int(n_pos), "pages_out": int(n_pages),
           "device_ms": round(tot_ms, 3), "queries_per_s": round(nq / (tot_ms / 1e3), 0) if tot_ms else 0.0,
           # wall time of the dcd_query_batch calls themselves (host planning + device + the per-query counts coming back; the
           # CSR result arrays stay on the device: dcd_query_opts.no_download) -- what a host thread waits for
           "call_wall_ms": round(wall_ms, 3), "queries_per_s_wall": round(nq / (wall_ms / 1e3), 0) if wall_ms else 0.0,
           "gpostings_per_s": round(npost / 1e9 / (tot_ms / 1e3), 2) if tot_ms else 0.0,
           "roofline": {"bound": "hbm", "algorithmic_bytes": int(a_query), "achieved": round(a_query / 1e9 / (tot_ms / 1e3), 1) if tot_ms else 0.0,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a_query / 1e9 / (tot_ms / 1e3) / HBM_PEAK_GBS, 4) if tot_ms else 0.0,
                        "bytes": "2 * units of every operand list + 4 * positions + 8 * pages (SURVEY 8d), exact counts",
                        # throughput, not kernel credit: an operator that several queries of the batch name is evaluated once
                        "bytes_after_cse": int(a_cse),
                        "frac_after_cse": round(a_cse / 1e9 / (tot_ms / 1e3) / HBM_PEAK_GBS, 4) if tot_ms else 0.0},
           "common_subexpressions": {"operators_named": int(ops_named), "operators_evaluated": int(ops_eval),
                                     "operand_units_named": int(un_named), "operand_units_evaluated": int(un_eval)},
           "calls": len(per), "call_ms": per[:6], "label": label}
    return out, progs


def check_totals(leg, docs, res):
    """The seeded legs have known result totals (tests/golden/bench_query_totals.json): other totals mean that a bench-only path
    computes something else now -- the line would be worthless, so the run fails."""
    path = os.path.join(ROOT, "tests", "golden", "bench_query_totals.json")
    if not os.path.exists(path):
        return
    want = json.load(open(path))["legs"].get("%s:%d:%d" % (leg, docs, res["queries"]))
    if want is None:
        return
    got = {"pages_out": res["pages_out"], "positions_out": res["positions_out"]}
    if got != want:
        raise SystemExit("bench.py: leg %s on %d docs produced %s, the golden totals are %s" % (leg, docs, got, want))
    res["totals_match_golden"] = True


def and_queries(ranks, exact=False):
    from docodo_amd import synth
    out = []
    for row in ranks:
        ws = [synth.term(int(k)) for k in row]
        out.append([(w, -(len(w) + 4) if exact else 255 + len(w)) for w in ws])
    return out


def phrase_queries(n, n_docs, seed):
    """configs[4]: 2- (even query id) or 3-word (odd) phrases taken from actual corpus positions (>= 1 hit guaranteed),
    quoted => R = -(len + 4)."""
    from docodo_amd import synth
    rng = np.random.default_rng(seed)
    out = []
    for q in range(n):
        doc = int(rng.integers(0, n_docs))
        words = bytes(synth.corpus_host(1, first_doc=doc)).decode("ascii").split()
        k = 2 + (q & 1)
        if len(words) < k:
            continue
        s = int(rng.integers(0, len(words) - k + 1))
        out.append([(w, -(len(w) + 4)) for w in words[s:s + k]])
    return out


def oracle_query_sample(oix, progs, budget_s, min_q=8):
    """The oracle (1 thread: the reference serialises Search under one lock, Search.cs:448) on the first queries of the leg,
    until `budget_s` seconds are used."""
    done, t_used = 0, 0.0
    pa, oa = progs[0]
    k = min_q
    while done < oa.size - 1 and t_used < budget_s:
        k = min(k, oa.size - 1 - done)
        sub_off = oa[done:done + k + 1] - oa[done]
        t0 = time.perf_counter()
        oix.query_batch(pa[int(oa[done]):int(oa[done + k])], sub_off)
        t_used += time.perf_counter() - t0
        done += k
        k *= 2
    return {"queries_per_s": round(done / t_used, 2), "cores": 1, "kind": "port", "queries": done,
            "sample": "first %d queries of the leg on the same index bytes, oracle 1 thread, %.1f s" % (done, t_used)}


def oracle_query_all_cores(oix, progs, n_1core, budget_s):
    """the reference serialises Search under ONE lock (Search.cs:448; GetPage keeps a cursor in the page list,
    Build.cs:74-97), so queries/s of one index object is the one-thread figure.  This is the query-sharded variant on ALL host
    cores: the threads share the loaded index (posting lists and page table are only read), each with a page cursor of its own
    (oracle.cpp orc_query_batch_mt) -- what one private copy of the index per thread gives, without the copies.  The sample's
    queries in contiguous slices, wall time of the slowest thread."""
    C = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    pa, oa = progs[0]
    total = oa.size - 1
    if C < 2 or total < C:
        return None
    # slices of a few queries per thread until the time budget is used (the threads share the memory system: the speed-up over one
    # thread is far from C on an index whose page table is walked from its start by every query -- GetPage is a forward cursor)
    done, t_used, k = 0, 0.0, 2 * C
    while done < total and t_used < budget_s:
        k = min(k, total - done)
        t0 = time.perf_counter()
        oix.query_batch_mt(pa[int(oa[done]):int(oa[done + k])], oa[done:done + k + 1] - oa[done], C)
        t_used += time.perf_counter() - t0
        done += k
        k *= 2
    return {"queries_per_s": round(done / t_used, 2), "cores": C, "queries": done,
            "sample": "first %d queries of the leg, %d threads on one shared index (query-sharded, private page cursors), %.1f s" % (done, C, t_used)}


def query_bench(ix, n_single, n_and3, cpu_sample, img=None, lst=None, seed_offset=0):
    """configs[1] query leg (10 k single-term queries over ranks 101..100k, Appendix F) plus a batch of
    3-term AND queries (opt == null -> Dist = 255) on the 1 M-doc index."""
    from docodo_amd import synth
    info = TermInfo(ix)
    out = {}
    oix = None
    if cpu_sample and img is not None:
        import oracle
        oix = oracle.Index(stemmers=[])
        oix.load(img, lst)
    # the FIRST batch of a handle also makes the resident coordinates and the cell directories (adoption: two passes over the units)
    tw = time.perf_counter()
    adopt_dev = ix.prepare()
    ix.query_batch(np.array([0, 0, 258], np.int32), np.array([0, 3], np.uint64), totals=True)
    out["first_query_ready_ms"] = round((time.perf_counter() - tw) * 1e3, 2)
    out["adopt_device_ms"] = round(adopt_dev, 3)
    for name, nq, nt in (("single_term", n_single, 1), ("and3", n_and3, 3)):
        if nq <= 0:
            continue
        qs = and_queries(synth.query_ranks(nq, nt, seed=synth.SEED_QUERY + seed_offset))
        out[name], progs = run_query_leg(ix, qs, info, max(nq, 1), "configs[1] index", reps=3)
        out[name]["best_of"] = 3
        if seed_offset == 0:
            check_totals("configs1_" + name, int(ix.n_docs), out[name])
        if oix is not None and progs:
            out[name]["cpu_baseline"] = oracle_query_sample(oix, progs, cpu_sample)
            ac = oracle_query_all_cores(oix, progs, out[name]["cpu_baseline"]["queries"], cpu_sample)
            if ac:
                out[name]["cpu_baseline"]["all_cores"] = ac
    return out


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (this process has not touched HIP;
    torch.cuda.device_count() does not initialise it), forward rank 0's JSON line, exit with the child's code."""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not os.environ.get("DOCODO_BENCH_REHEARSAL"):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node (DOCODO_BENCH_REHEARSAL=1 rehearses the N-rank path on "
                         "one GPU over gloo -- plumbing only, not a measurement)" % (args.gpus, n_dev))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    for l in r.stdout.splitlines():
        if not l.startswith('{"metric"'):
            print(l, file=sys.stderr)
    if r.returncode == 0 and len(lines) != 1:
        raise SystemExit("bench.py: the %d ranks ended without the result line" % args.gpus)
    if lines:
        print(lines[-1], flush=True)
    sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs", type=int, default=1_000_000, help="documents per GPU (1 KiB each) of the timed configs[1] build")
    ap.add_argument("--cpu-docs", type=int, default=250_000, help="documents in the 1-core CPU-baseline sample (0 = skip every CPU leg)")
    ap.add_argument("--cpu-docs-all", type=int, default=1_000_000, help="documents in the all-cores CPU-baseline sample (0 = skip)")
    ap.add_argument("--queries", type=int, default=10_000, help="single-term queries of configs[1] (0 = skip the query leg)")
    ap.add_argument("--and3", type=int, default=20_000, help="3-term AND queries in the extra query batch on the configs[1] index")
    ap.add_argument("--total-docs", type=int, default=10_000_000, help="configs[2..4]: documents of the big index (strong scaling: total-docs / N "
                    "per rank); 0 = skip those legs")
    ap.add_argument("--c4-queries", type=int, default=100_000, help="configs[3]: 3-term AND queries over the big index")
    ap.add_argument("--c5-queries", type=int, default=100_000, help="configs[4]: exact-phrase queries over the big index")
    ap.add_argument("--query-chunk", type=int, default=0, help="queries per dcd_query_batch call in the configs[3] / [4] legs (0 = the whole leg in ONE call)")
    ap.add_argument("--inflight", type=int, default=2, help="extra (untimed for `value`) leg: this many builders at once, as the "
                    "reference's MaxDegreeOfParallelism index tasks (Index.cs:127, 464-465); 0/1 = skip")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("DOCODO_BENCH_PIPELINE", "1")),
                    help="N > 1: 1 = the exchange + merge of step i runs on a second host thread beside the build of step i + 1 "
                    "(every merge completes inside the timed region); 0 = one after the other")
    ap.add_argument("--lean-events", type=int, default=0, help="1: the timed steps record HIP events around the dominant kernel only "
                    "(chosen in the fully instrumented warm-up steps: +1-2 %% value); 0 (default): every event in the timed steps -- the "
                    "default keeps the per-kernel and per-stage figures of the line measured in the timed region itself")
    ap.add_argument("--host-legs", type=int, default=1, help="1: also time the UTF-16 build and the build from pinned host memory (extra legs of the line)")
    ap.add_argument("--force-exchange", action="store_true", help="rehearsal: run the N > 1 step (exchange + merge) with one rank")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args, sys.argv[1:])

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner to fd 1 when a communicator
    # is created): from here on fd 1 is stderr, and the result line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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
        if "MASTER_PORT" not in os.environ:            # (one rank, no launcher: any free port)
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = "cuda" if backend == "nccl" else "cpu"      # where the small reductions of this script live

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
    # the library's own communicator (dcd_comm_create: RCCL over xGMI; the rehearsal on one GPU: the host transport).  torch.distributed
    # only carries rank 0's id here, and the barriers / reductions of this script.
    comm = ddist.Comm.from_torch(local_rank) if exchange else None

    def merge(b):
        # exchange step, inside the library (dcd_build_merge_distributed): per-term segments travel to their owner (heavy terms:
        # balanced by postings; the rest: weighted hash slices) with ncclSend / ncclRecv groups and are merged there;
        # part of the timed step
        t0 = time.perf_counter()
        owned = ddist.distributed_merge(b, comm)
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
    note("warm-up done, timing %d steps" % args.steps)
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
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    out = None
    if rank == 0:
        N, V, U, img_bytes, passes, id_keys, hot_terms, hot_post = last
        dom_timed = (ksum.copy(), kcnt.copy())
        if lean:                           # the report's per-kernel / per-stage figures: the fully instrumented warm-up steps
            ksum, kcnt, stage = cal_k * args.steps / cal_steps, cal_n * args.steps / cal_steps, cal_stage * args.steps / cal_steps
        lps = kcnt / args.steps            # launches per step of every kernel

        def abytes_of(k):
            hot32 = id_keys == 2 and hot_terms > 0 and os.environ.get("DOCODO_BUILD_HOT32", "1") != "0"    # (the library's own rule)
            return algorithmic_bytes(KERNEL_NAMES[k], T, N, U, lps[k], hot_post, bool(id_keys), passes, hot32)
        ms_step = dt / args.steps * 1e3
        value = world * T / 1e9 / (dt / args.steps)
        # dominant kernel = largest total device time per step (HIP events on the build stream)
        per_step = ksum / args.steps
        dom = int(np.argmax(per_step[:7]))
        name = KERNEL_NAMES[dom]
        avg_ms = dom_timed[0][dom] / max(1.0, dom_timed[1][dom])      # (the timed region's own launches of the dominant kernel)
        abytes = abytes_of(dom)
        achieved = abytes / 1e9 / (avg_ms / 1e3)
        # the whole build on the SURVEY 8(d) bytes: text + final posting units + one u64 per page, over the wall time of a step
        a_build = T + 2 * U + 8 * D
        build_alg = a_build / 1e9 / (dt / args.steps)
        workload = ("configs[1]: %d synthetic 1 KiB docs per GPU, Zipf-1.1 over 100k terms (SURVEY App. F); " % D) + (
            "doc-range shard build (shard-only: no .index image / term-ordered copy per shard) + RCCL exchange + per-owner merge of the per-term "
            "segments, text resident in HBM -> hash-partitioned merged posting lists resident in HBM" if exchange else
            "full index build, text resident in HBM -> .index image resident in HBM")
        out = {
            "metric": "text_indexed_GB_per_s", "value": round(value, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "shard_only": bool(exchange),
                       "docs_per_gpu": D, "text_bytes_per_gpu": T, "postings": int(N), "terms": int(V), "units": int(U),
                       "index_bytes": int(img_bytes), "sort_passes": int(passes), "sort": (
                           ("by the term ids the vocabulary table hands out while the tokeniser fills it (tuples carry their sort keys); "
                            "ordinal vocabulary sort on a second stream" + ("; %d hot terms chosen by a sampling launch (%.1f %% of the postings) "
                            "final after the first radix pass" % (hot_terms, 100.0 * hot_post / max(1, N)) if hot_terms else "")) if id_keys == 2 else
                           "by dense vocabulary id, ordinal vocabulary sort on a second stream; %d hot terms (%.1f %% of the postings) final "
                           "after the first radix pass" % (hot_terms, 100.0 * hot_post / max(1, N))) if id_keys else "by ordinal rank",
                       "hot_postings": int(hot_post),
                       "sharding": "doc-range, global coordinates", "world_size": world,
                       "backend": (dist.get_backend() if dist is not None else None)},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_copy": round(achieved / HBM_COPY_GBS, 4),
                         "algorithmic_bytes_per_launch": int(abytes), "avg_launch_ms": round(avg_ms, 4),
                         "launches_per_step": round(kcnt[dom] / args.steps, 2), "traffic": measured_traffic(name, T, int(N), int(U)),
                         "limiter": LIMITER.get(name, "hbm"),
                         # `frac` above prices the DOMINANT KERNEL on its own algorithmic bytes; the figure below prices the WHOLE
                         # BUILD on the bytes SURVEY 8(d) defines (T + 2 U + 8 P per GPU) over the wall time of a step
                         "frac_build_algorithmic": round(build_alg / HBM_PEAK_GBS, 4),
                         "build_algorithmic": {"bytes_per_step_per_gpu": int(a_build), "achieved": round(build_alg, 1), "unit": "GB/s per GPU",
                                               "frac": round(build_alg / HBM_PEAK_GBS, 4), "frac_of_measured_copy": round(build_alg / HBM_COPY_GBS, 4),
                                               "bytes": "T + 2 U + 8 P (text + final posting units + one u64 per page, SURVEY 8d)"}},
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
            out["config"]["exchange"] = ("RCCL all-to-all of per-term segments (meta, keys, units): heavy terms to owners balanced by posting units (one small "
                                         "all-gather), the others to hash(key) % world; dcd_merge_begin overlaps the unit exchange, then dcd_merge_finish" +
                                         ("; the exchange + merge of step i runs on a second host thread beside the build of step i + 1, all of them inside the timed region"
                                          if worker is not None else ""))
            out["config"]["scaling_note"] = ("the N = 1 line builds the .index image per step; shard-only steps skip it (k_img_units: ~8 % of a step) and "
                                             "add the exchange + merge instead")
        note("timed steps done: %.3f ms per step" % ms_step)
        if world == 1 and not exchange:
            out["general_pages"] = general_pages_leg(text, D, T, local_rank, max(3, args.steps // 2))
        if world == 1 and not exchange and args.host_legs:
            # the inputs a C# host really hands over: a `string` (UTF-16 code units) and host memory (PCIe in the timed region)
            out["utf16"] = utf16_leg(text, D, local_rank, max(3, args.steps // 2))
            out["pcie"] = pcie_leg(D, local_rank, 3)
        if world == 1 and args.inflight > 1 and not exchange:
            out["builders_in_flight"] = inflight_bench(args.inflight, args.steps, text, D, local_rank, T)
        if world == 1 and args.cpu_docs > 0 and not exchange:
            note("cpu baseline (oracle) ...")
            out["cpu_baseline"] = cpu_baseline(min(args.cpu_docs, D), args.cpu_docs_all)
            note("cpu baseline done")
        if world == 1 and args.queries > 0 and not exchange:
            # query leg of configs[1], outside the timed build steps
            bq = docodo_amd.Builder(coord_base=coord_base, device=local_rank)
            bq.AddUniformPagesDev(text.data_ptr(), D, synth.DOC_BYTES, first_doc=first_doc)
            bq.Finish()
            img, lst = (bq.index_image(), bq.list_image()) if args.cpu_docs > 0 else (None, None)
            ixq = bq.Build()
            out["queries"] = query_bench(ixq, args.queries, args.and3, 6.0 if args.cpu_docs > 0 else 0, img, lst)
            note("configs[1] query leg done")
            ixq.close()
            bq.close()
            del img, lst
    del text
    torch.cuda.empty_cache()
    docodo_amd._lib.pool_trim(local_rank)
    if args.total_docs > 0 and not args.force_exchange:
        big = big_index_legs(args, rank, world, local_rank, dist, cdev, comm)
        if rank == 0:
            out["configs234"] = big
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def general_pages_leg(text, D, T, device, steps):
    """The build a real host drives: dcd_build_add_pages_dev with EXPLICIT page offsets (one per page, host array) on the same
    resident text.  Wall time per build (add + finish), device time of the finish."""
    import docodo_amd
    from docodo_amd import synth
    off = np.arange(D + 1, dtype=np.uint64) * synth.DOC_BYTES
    ms_dev, walls = [], []
    for i in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = docodo_amd.Builder(device=device)
        b.AddPagesDev(text.data_ptr(), T, off, docs_per_page=True)
        st = b.Finish()
        torch.cuda.synchronize()
        if i:
            walls.append(time.perf_counter() - t0)
            ms_dev.append(st.ms_total)
        b.close()
    w = float(np.mean(walls))
    return {"GB_per_s": round(T / 1e9 / w, 2), "ms_per_build_wall": round(w * 1e3, 3), "ms_per_build_device": round(float(np.mean(ms_dev)), 3),
            "builds": steps, "note": "explicit page offsets (dcd_build_add_pages_dev, host offset array); `value` is the uniform-pages entry point"}


def utf16_leg(text, D, device, steps):
    """The same corpus as UTF-16LE code units (what a C# `string` pins to; dcd_build_opts.encoding = DCD_ENC_UTF16LE), resident in
    HBM: characters / s and bytes / s of the 2-byte text.  Never `value`."""
    import docodo_amd
    from docodo_amd import synth
    t16 = text.to(torch.int16)
    torch.cuda.synchronize()
    ms_dev, walls = [], []
    for i in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE, device=device)
        b.AddUniformPagesDev(t16.data_ptr(), D, synth.DOC_BYTES)
        st = b.Finish()
        torch.cuda.synchronize()
        if i:
            walls.append(time.perf_counter() - t0)
            ms_dev.append(st.ms_total)
        b.close()
    del t16
    w = float(np.mean(walls))
    n_chars = D * synth.DOC_BYTES
    return {"Gchars_per_s": round(n_chars / 1e9 / w, 2), "GB_per_s_of_utf16_bytes": round(2 * n_chars / 1e9 / w, 2),
            "ms_per_build_wall": round(w * 1e3, 3), "ms_per_build_device": round(float(np.mean(ms_dev)), 3), "builds": steps,
            "note": "extra: the configs[1] corpus as UTF-16LE (a C# string), resident; `value` is the 1-byte text"}


def pcie_leg(D, device, steps):
    """The text starts in PINNED HOST memory and crosses PCIe inside the timed region: dcd_build_add_pages (what a C# host calls with a
    pinned string / byte[]) + finish, end to end.  Never `value` (the contract's value has its input resident in HBM)."""
    import docodo_amd
    from docodo_amd import synth
    text = torch.from_numpy(synth.corpus_host(D)).pin_memory().numpy()
    off = np.arange(D + 1, dtype=np.uint64) * synth.DOC_BYTES
    add_ms, fin_ms = [], []
    for i in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = docodo_amd.Builder(device=device)
        b.AddPages(text, off, docs_per_page=True)
        t1 = time.perf_counter()
        b.Finish()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        b.close()
        if i:
            add_ms.append((t1 - t0) * 1e3)
            fin_ms.append((t2 - t1) * 1e3)
    tot = float(np.mean(add_ms)) + float(np.mean(fin_ms))
    return {"GB_per_s": round(D * synth.DOC_BYTES / 1e9 / (tot / 1e3), 2), "add_pages_ms": round(float(np.mean(add_ms)), 2),
            "finish_ms": round(float(np.mean(fin_ms)), 2), "builds": steps,
            "note": "extra: host-resident (pinned) text through dcd_build_add_pages, H2D copy inside the timed region"}


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


def big_index_legs(args, rank, world, device, dist, cdev, comm=None):
    """configs[2..4].  Build: total_docs documents, doc-range sharded over the ranks (strong scaling; a rank's range is cut
    into segments of < 2^32 characters), global coordinates, merged on the device -- N = 1: dcd_merge_shards over the
    segments; N > 1: exchange to the term owners + merge (the hash-partitioned index), then the replica all-gather, so that
    every GPU holds the complete index.  Queries: the batch is split evenly by query id over the ranks (no exchange), every
    rank evaluates its share on its replica; aggregate = all queries / slowest rank."""
    import docodo_amd
    from docodo_amd import dist as ddist, synth
    total = args.total_docs
    per = -(-total // world)
    lo, hi = min(total, rank * per), min(total, (rank + 1) * per)
    n_mine = hi - lo
    d = torch.empty(max(1, n_mine) * synth.DOC_BYTES, dtype=torch.uint8, device="cuda")
    synth.corpus_dev(d.data_ptr(), n_mine, first_doc=lo, device=device)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def build_once():
        sb = ddist.SegmentedBuilder(synth.DOC_BYTES, device=device, first_page=lo, defer_export=world > 1)
        sb.add_pages_dev(d.data_ptr(), n_mine)
        n_seg, dev_ms, n_post = sb.n_segments, sb.device_ms, sb.n_postings
        if world == 1:
            ix = sb.finish()
        else:
            owned = sb.finish_distributed(comm)         # this rank's doc range -> its hash partition of the global index
            ix = ddist.allgather_index(owned, comm)
            owned.close()
            ddist.set_uniform_pages(ix, total, synth.DOC_BYTES)
        return ix, n_seg, dev_ms, n_post

    note("configs[2]: building the %d-doc index" % total)
    ix, n_seg, dev_ms, n_post = build_once()            # warm-up (allocates)
    ix.close()
    barrier()
    t0 = time.perf_counter()
    ix, n_seg, dev_ms, n_post = build_once()
    barrier()
    dt = time.perf_counter() - t0
    del d
    torch.cuda.empty_cache()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        pp = torch.tensor([float(n_post)], dtype=torch.float64, device=cdev)
        dist.all_reduce(pp, op=dist.ReduceOp.SUM)
        n_post = int(pp.item())
    out = {"index": {"docs": total, "text_bytes": total * synth.DOC_BYTES, "postings": int(n_post), "terms": ix.Count,
                     "segments_per_rank": n_seg, "scaling": "strong", "n_gpus": world,
                     "build_wall_ms": round(dt * 1e3, 2), "shard_build_device_ms_rank0": round(dev_ms, 2),
                     "text_GB_per_s": round(total * synth.DOC_BYTES / 1e9 / dt, 2),
                     "pipeline": ("segments (shard-only) -> dcd_merge_shards" if world == 1 else
                                  "segments -> per-rank shard index -> RCCL exchange to balanced owners + merge -> replica all-gather + merge"),
                     "note": "configs[2]: whole-job wall time from resident text to the merged index on every GPU"}}
    note("configs[2] index built: %.1f ms" % (dt * 1e3))
    # ---- configs[3] / configs[4] ----
    info = TermInfo(ix)
    torch.cuda.synchronize()
    tw = time.perf_counter()
    adopt_dev = ix.prepare()                                                                   # adoption: resident coordinates + cell directories
    ix.query_batch(np.array([0, 0, 258], np.int32), np.array([0, 3], np.uint64), totals=True)
    out["index"]["first_query_ready_ms"] = round((time.perf_counter() - tw) * 1e3, 2)          # (wall: incl. the hipMalloc of the 8 N + cell bytes)
    out["index"]["adopt_device_ms"] = round(adopt_dev, 3)
    out["index"]["note"] = ("configs[2]: build_wall_ms = whole-job wall time from resident text to the merged index on every GPU; the first query batch "
                            "then needs first_query_ready_ms more (resident coordinates + cell directories of the query side)")
    legs = []
    if args.c4_queries > 0:
        legs.append(("c4_and3", "configs[3]: 3-term AND, Dist = 255, terms ~ Zipf over ranks 101..100k", and_queries(synth.query_ranks(args.c4_queries, 3)), 4.0))
    if args.c5_queries > 0:
        legs.append(("c5_phrase", "configs[4]: exact 2- / 3-word phrases sampled from corpus positions", phrase_queries(args.c5_queries, total, synth.SEED_QUERY), 12.0))
    oix = None
    note("query sets ready")
    for name, label, qs, budget in legs:
        mine = qs[rank::world]                          # sharded by query id
        # ONE dcd_query_batch call per leg (round 4): the library cuts it into chunks itself, plans chunk k + 1 on the host while
        # chunk k runs, and shares the operators that several queries name
        res, progs = run_query_leg(ix, mine, info, args.query_chunk if args.query_chunk > 0 else max(1, len(mine)), label)
        note("%s: %d queries, %.1f ms device" % (name, res["queries"], res["device_ms"]))
        if world == 1:
            check_totals(name, total, res)
        if dist is not None:
            v = torch.tensor([res["queries"], res["postings_in"], res["units_in"], res["positions_out"], res["pages_out"]], dtype=torch.float64, device=cdev)
            m = torch.tensor([res["device_ms"]], dtype=torch.float64, device=cdev)
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            ms = float(m.item())
            nq, npost, nun, npos, npg = (int(x) for x in v.tolist())
            a_q = 2 * nun + 4 * npos + 8 * npg
            res.update({"queries": nq, "postings_in": npost, "units_in": nun, "positions_out": npos, "pages_out": npg,
                        "device_ms": round(ms, 3), "queries_per_s": round(nq / (ms / 1e3), 0), "gpostings_per_s": round(npost / 1e9 / (ms / 1e3), 2),
                        "sharding": "by query id over %d replicas; device time of the slowest rank" % world})
            res["roofline"].update({"algorithmic_bytes": a_q, "achieved": round(a_q / 1e9 / (ms / 1e3), 1),
                                    "frac": round(a_q / 1e9 / (ms / 1e3) / (HBM_PEAK_GBS * world), 4), "peak": HBM_PEAK_GBS * world})
        if name == "c4_and3":
            # HBM-side bytes of the leg's dominant kernel from the committed PMC pass (valid for the configuration it was measured on)
            tp = os.path.join(ROOT, "profiles", "r04_c4_traffic.json")
            if os.path.exists(tp):
                tr = json.load(open(tp))
                if tr.get("total_docs") == args.total_docs and tr.get("queries"):
                    res["roofline"]["traffic"] = int(tr["fetch_bytes"] / tr["queries"] * res["queries"])
                    res["roofline"]["traffic_note"] = ("k_pair_parts + k_pair_stream, FETCH_SIZE per query of a 20 k-query call (profiles/r04_c4_traffic.json) "
                                                       "x the leg's queries: the operands at 8 B per coordinate + 4 B per directory cell")
        if rank == 0 and world == 1 and args.cpu_docs > 0 and progs:
            if oix is None:
                import oracle
                oix = oracle.Index(stemmers=[])
                big_img, big_lst = ix.index_image(), ix.list_image()
                oix.load(big_img, big_lst)
                note("oracle loaded the big index")
            res["cpu_baseline"] = oracle_query_sample(oix, progs, budget, min_q=2)
            ac = oracle_query_all_cores(oix, progs, res["cpu_baseline"]["queries"], budget)
            if ac:
                res["cpu_baseline"]["all_cores"] = ac
            note("%s: oracle samples done" % name)
        out[name] = res
    ix.close()
    return out


if __name__ == "__main__":
    main()
