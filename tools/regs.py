"""Register / scratch / occupancy table of the kernels in one csrc/*.hip (cross-compiled to gfx950 assembly with the
Makefile's flags for that file).  usage: python tools/regs.py nnconv_mfma [extra hipcc flags]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gnn_qot_estimation_amd", "csrc")
name = sys.argv[1]
mk = open(os.path.join(src, "Makefile")).read()
var = dict(re.findall(r"^(\w+)\s*=\s*(.*)$", mk, re.M))
flags = var.get("FLAGS_" + name, "")
flags = re.sub(r"\$\((\w+)\)", lambda m: var.get(m.group(1), ""), flags)
out = f"/tmp/{name}.s"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{root}/include", f"-I{src}", "-Wno-unused-result",
       "-Wno-pass-failed", "-S", "--cuda-device-only", "-o", out, os.path.join(src, name + ".hip")] + flags.split() + sys.argv[2:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
for f in re.split(r"\n(?=_Z\w+:)", txt):
    m = re.match(r"(_Z\w+):", f)
    if not m or "NumVgprs" not in f:
        continue
    g = lambda k: re.search(r"; %s: (\d+)" % k, f).group(1)
    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    print(f"{dem.split('(')[0][:70]:70s} vgpr {g('NumVgprs'):>3s} agpr {g('NumAgprs'):>3s} scratch {g('ScratchSize'):>4s} occ {g('Occupancy')}"
          f" mfma {f.count('v_mfma')}")
print("asm:", out)
