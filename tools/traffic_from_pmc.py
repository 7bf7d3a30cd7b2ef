"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the eager step.

usage: python tools/traffic_from_pmc.py <FETCH csv> <WRITE csv> [out json]
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: counters are in KiB-ish units of 1024 B and
gfx950 reports FETCH_SIZE in 2x units (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import collections, csv, json, sys

NAMES = {                      # kernel-name fragment -> bench.py kernel_table key
    "nnconv_adjoint_dw64_kernel": "nnconv_adjoint_dw",
    "nnconv_mfma64_kernel": "nnconv_fused_fwd",
    "nnconv_gradh64_kernel": "nnconv_gradh_fused",
    "tconv_fwd_kernel": "tconv_fwd",
    "tconv_fwd_tile_kernel": "tconv_fwd_tile",
    "roles_kernel": "roles (prologue / epilogue launches, mean)",
    "tconv_bwd_dst_kernel": "tconv_bwd_dst",
    "tconv_bwd_src_kernel": "tconv_bwd_src",
}


def per_launch(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        for frag, key in NAMES.items():
            if frag in name:
                agg[key][0] += 1
                agg[key][1] += float(r["Counter_Value"])
    return {k: v / n for k, (n, v) in agg.items()}


fetch, write = per_launch(sys.argv[1]), per_launch(sys.argv[2])
out = {k: int((2 * fetch[k] + write.get(k, 0.0)) * 1024) for k in fetch}
out["_note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of the "
                "eager step (tools/pmc.sh), gfx950 FETCH_SIZE x2 correction per MI355X_MICROARCH.md; cfg2 B=1024: "
                "N=102400, E=409600, H=64; TransformerConv in table mode; sources: " + " ".join(sys.argv[1:3]))
json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
