"""Per-kernel sums of rocprofv3 --pmc results (rocpd sqlite databases OUT/g*/p_results.db)."""
import collections, glob, os, sqlite3, sys
out = sys.argv[1]
want = sys.argv[2:] or ["k_rs_scatter", "k_rs_hist", "k_tok_emit", "k_tok_count", "k_enc_write", "k_enc_reduce"]
tab = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "g*", "*.db"))):
    db = sqlite3.connect(f)
    for name, cname, val, d in db.execute("select name, counter_name, counter_value, duration from pmc_events"):
        for w in want:
            if w in name:
                tab[w][cname] += val
                cnt[w][cname] += 1
for w in want:
    if w not in tab: continue
    print("==", w)
    for c in sorted(tab[w]):
        print("  %-32s %16.0f  (%d dispatches, %.0f per dispatch)" % (c, tab[w][c], cnt[w][c], tab[w][c] / max(1, cnt[w][c])))
