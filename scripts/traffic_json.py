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
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on python3 bench.py --steps 3 --warmup 1 "
                 "--cpu-docs 0 --queries 0; values are the per-launch maxima (KB * 1024). On gfx950 FETCH_SIZE counts exactly half of a "
                 "wide coalesced 16 B/lane stream (MI355X_MICROARCH.md, HBM): fetch_bytes_x2 applies that correction to the whole read "
                 "side (upper bound for kernels that also gather).",
       "T": b["config"]["text_bytes_per_gpu"], "N": b["config"]["postings"], "U": b["config"]["units"], "kernels": {}}
for k in sorted(fetch):
    f, w = fetch[k], write.get(k, 0.0)
    out["kernels"][k] = {"fetch_size_kb_max": f, "write_size_kb_max": w, "launches_seen": seen[k], "fetch_bytes_raw": int(f * 1024),
                         "write_bytes": int(w * 1024), "fetch_bytes_x2": int(2 * f * 1024)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("wrote", sys.argv[4], len(out["kernels"]), "kernels")
