"""MFMA / VALU busy fractions per kernel from the JSON tools/pmc_multi.sh leaves (counters SQ_VALU_MFMA_BUSY_CYCLES,
SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE; one rocprofv3 --pmc pass).  usage: python tools/busy_from_pmc.py <in json> <out json>"""
import json, sys
d = json.load(open(sys.argv[1]))
out = {}
for k, c in d.items():
    gui = c.get("GRBM_GUI_ACTIVE")
    if not gui:
        continue
    simd_cycles = gui / 8 * 1024
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
    va = 4 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles
    out[k] = {"mfma_busy": round(mf, 3), "valu_busy": round(va, 3), "gui_active_cycles": gui, "idle": round(1 - mf - va, 3)}
out["_note"] = ("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); VALU busy = 4 * SQ_ACTIVE_INST_VALU / same; "
                "one rocprofv3 --pmc pass of the eager cfg2 step (tools/pmc_multi.sh); source " + sys.argv[1])
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
