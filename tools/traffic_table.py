"""Per-kernel HBM traffic and achieved bandwidth from THREE rocprofv3 runs of one command: --pmc FETCH_SIZE, --pmc WRITE_SIZE
(separate passes, `counter_collection.csv`) and --kernel-trace --stats (`kernel_stats.csv`).

usage: python tools/traffic_table.py <FETCH csv> <WRITE csv> <kernel_stats csv> <out json> [note]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (the counters' unit is 1024 B; gfx950 reports FETCH_SIZE at half
the bytes of a wide streaming read: MI355X_MICROARCH.md, HBM section).  Kernels are keyed by their name up to the argument
list (template arguments kept)."""
import collections, csv, json, sys


def key(name):
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        if ch == "(" and depth == 0:
            break
        out.append(ch)
        if ch == ">":
            depth -= 1
    k = "".join(out).strip()
    for pre in ("void ", "qot::"):
        k = k.replace(pre, "")
    return k


def per_launch(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = key(r.get("Kernel_Name", ""))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: (v / n, n) for k, (n, v) in agg.items()}


fetch, write = per_launch(sys.argv[1]), per_launch(sys.argv[2])
stats = {key(r["Name"]): (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["Percentage"])) for r in csv.DictReader(open(sys.argv[3]))}
rows = []
for k, (f, n) in fetch.items():
    w = write.get(k, (0.0, 0))[0]
    us, calls, pct = stats.get(k, (None, None, None))
    b = (2 * f + w) * 1024
    rows.append({"kernel": k, "launches_in_pmc_pass": n, "hbm_read_bytes": int(2 * f * 1024), "hbm_write_bytes": int(w * 1024),
                 "hbm_bytes_per_launch": int(b), "avg_us": None if us is None else round(us, 1),
                 "GBps": None if not us else round(b / us / 1e3, 1), "share_of_kernel_time_pct": pct})
rows.sort(key=lambda r: -(r["share_of_kernel_time_pct"] or 0))
out = {"_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, separate rocprofv3 --pmc passes; avg_us from the "
                "--kernel-trace --stats run of the same command; " + (sys.argv[5] if len(sys.argv) > 5 else ""),
       "sources": sys.argv[1:4], "kernels": rows[:40]}
json.dump(out, open(sys.argv[4], "w"), indent=1)
for r in rows[:24]:
    print(f"{r['kernel'][:70]:70s} {r['hbm_bytes_per_launch'] / 1e6:10.1f} MB {str(r['avg_us']):>9s} us {str(r['GBps']):>8s} GB/s {str(r['share_of_kernel_time_pct']):>6s}%")
