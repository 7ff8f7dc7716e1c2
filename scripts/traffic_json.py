"""Builds profiles/<tag>_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: traffic_json.py FETCH_counter_collection.csv WRITE_counter_collection.csv bench.json out.json"""
import csv, json, re, sys, collections

def per_kernel(path, counter):
    mx, seen = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", n)
        if not m:
            continue
        k = m.group(1)
        mx[k] = max(mx[k], float(r["Counter_Value"]))
        seen[k] += 1
    return mx, seen

fetch, seen = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
b = json.load(open(sys.argv[3]))
T, N, U = b["config"]["text_bytes_per_gpu"], b["config"]["postings"], b["config"]["units"]
NC = N - b["config"].get("hot_postings", 0)          # tuples of the cold passes
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on python3 bench.py --steps 3 --warmup 1 "
                 "--cpu-docs 0 --queries 0 --inflight 0; values are the per-launch maxima (KB * 1024). On gfx950 FETCH_SIZE counts a "
                 "streamed read at exactly half its bytes (MI355X_MICROARCH.md, HBM; here also for 8 B/lane streams: k_pack_write reads "
                 "8 N bytes and reports 4 N). traffic_bytes = write + fetch with the STREAMED part of the kernel's reads "
                 "(stream_read_bytes, its algorithmic read bytes) doubled and whatever it fetches beyond that (table gathers) taken as "
                 "counted; fetch_bytes_x2 doubles everything (upper bound).",
       "T": T, "N": N, "U": U, "kernels": {}}
# bytes every kernel reads as a stream (largest launch of the name)
STREAM = [("k_tok_emit", T), ("k_tok_count", T), ("k_rs_hist_strided", 8 * N), ("k_rs_scatter<8, true, true>", 8 * N),
          ("k_rs_scatter<8, false, true>", 8 * N), ("k_rs_scatter<9, false, true>", 8 * N),      # (strided = the first pass: all tuples)
          ("k_rs_scatter<9, true, true>", 8 * N), ("k_rs_scatter<10, true, true>", 8 * N),
          ("k_rs_hist", 8 * (NC if NC != N else N)), ("k_rs_scatter", 8 * (NC if NC != N else N)),
          ("k_pack_count", 8 * N), ("k_pack_write", 8 * N), ("k_img_units", 2 * U)]
for k in sorted(fetch):
    f, w = fetch[k], write.get(k, 0.0)
    stream = next((v for p, v in STREAM if k.startswith(p)), 0)
    raw = int(f * 1024)
    streamed_counted = min(raw, stream // 2)
    out["kernels"][k] = {"fetch_size_kb_max": f, "write_size_kb_max": w, "launches_seen": seen[k], "fetch_bytes_raw": raw,
                         "write_bytes": int(w * 1024), "fetch_bytes_x2": int(2 * f * 1024), "stream_read_bytes": int(stream),
                         "traffic_bytes": int(w * 1024) + raw + streamed_counted}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("wrote", sys.argv[4], len(out["kernels"]), "kernels")
