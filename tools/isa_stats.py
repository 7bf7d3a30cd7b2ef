"""ISA-level sanity numbers for the MFMA kernels: registers, scratch, and how many MFMAs sit directly behind a
full `s_waitcnt ...cnt(0)` (i.e. wait for their own operand load).  usage: python tools/isa_stats.py [file.hip]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gnn_qot_estimation_amd", "csrc", "nnconv_mfma.hip")
tmp = tempfile.mkdtemp()
asm = os.path.join(tmp, "k.s")
r = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"), "-Wno-pass-failed",
                    "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-S", "-o", asm, src],
                   capture_output=True, text=True, cwd=os.path.dirname(src))
res = {}
for b in r.stderr.split("Function Name: ")[1:]:
    name = b.split("\n")[0].split()[0]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    res[name] = (g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"))
lines = open(asm).read().split("\n")
for name, (vg, sc, lds) in res.items():
    if "ILi4" not in name and "Li64ELi4" not in name:
        continue
    best = None
    for i, l in enumerate(lines):
        if l.startswith(name + ":"):
            end = next((k for k in range(i, min(len(lines), i + 30000)) if "s_endpgm" in lines[k]), None)
            if end:
                b = lines[i:end]
                if best is None or sum("v_mfma" in x for x in b) > sum("v_mfma" in x for x in best):
                    best = b
    if not best:
        continue
    b = best
    mf = [i for i, l in enumerate(b) if "v_mfma" in l]
    if not mf:
        continue
    stall = 0
    for i in mf:
        j = i - 1
        while b[j].strip().startswith(";") or not b[j].strip():
            j -= 1
        if "s_waitcnt" in b[j] and ("lgkmcnt(0)" in b[j] or "vmcnt(0)" in b[j]):
            stall += 1
    mid = sum(1 for i, l in enumerate(b) if "scratch_" in l and mf[0] < i < mf[-1])
    print(f"{name[:58]:58s} VGPR {vg:3d} scratch {sc:3d} B LDS {lds:6d}  mfma {len(mf):3d}  behind waitcnt(0) {stall:3d}  "
          f"scratch ops between mfmas {mid}")
