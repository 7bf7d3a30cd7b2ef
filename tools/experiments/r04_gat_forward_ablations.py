"""Ablation builds of gat_fwd_kernel behind DESIGN 4.8's table (end of round 4).  Run in the container: every variant patches
csrc/gat.hip, builds the library, copies it to tools/diag/libqot_gnn_<name>.so and restores the source; then on ONE GPU box

    bash tools/prof_cfg.sh base cfg3 && for v in x1 x2 x3 x4 x5; do
        QOT_LIB_PATH=$GRAFT_REPO_ROOT/tools/diag/libqot_gnn_$v.so bash tools/prof_cfg.sh abl_$v cfg3; done

(the walk kernels repeat to 0.3 % on one box and vary by +-4 % between boxes: only same-box figures compare).  The variants
compute wrong results on purpose -- timing only."""
import os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "gnn_qot_estimation_amd", "csrc", "gat.hip")
base = open(src).read()
a = base.index("void gat_fwd_kernel(")
b = base.index("    if (bn_partials) {          // fixed order")


def variant(name, edits):
    k = base[a:b]
    for x, y in edits:
        assert k.count(x) >= 1, (name, x[:50])
        k = k.replace(x, y)
    open(src, "w").write(base[:a] + k + base[b:])
    subprocess.run(["make", "-C", os.path.dirname(src)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    os.makedirs(os.path.join(root, "tools", "diag"), exist_ok=True)
    shutil.copy(os.path.join(root, "gnn_qot_estimation_amd", "libqot_gnn.so"), os.path.join(root, "tools", "diag", f"libqot_gnn_{name}.so"))
    print("built", name)


try:
    # x1: no store of the output rows; x2: no exp; x3: no writes of the slot image; x4: no BatchNorm column sums; x5: no (max, denominator) store
    variant("x1", [("st4(out + i * G::HC + c, add4(d, ld4(bias + c)));", "if (d.x == 123456.f) st4(out + i * G::HC + c, add4(d, ld4(bias + c)));")])
    variant("x2", [("const float sc = __expf(m[v] - mn);", "const float sc = (m[v] - mn) * 0.5f;"),
                   ("const float pe = __expf(s - mn);", "const float pe = (s - mn) * 0.25f;")])
    variant("x3", [("mine[e * HC4 + sub + G::TPR * v] = zz[e][v];", "if (zz[e][v].x == 123456.f) mine[e * HC4 + sub + G::TPR * v] = zz[e][v];"),
                   ("lcache[(e * G::NV + v) * 256 + threadIdx.x] = as_[e][v];",
                    "if (as_[e][v] == 123456.f) lcache[(e * G::NV + v) * 256 + threadIdx.x] = as_[e][v];")])
    variant("x4", [("s1[v] = add4(s1[v], e);", ""),
                   ("s2[v] = make_float4(fmaf(e.x, e.x, s2[v].x), fmaf(e.y, e.y, s2[v].y), fmaf(e.z, e.z, s2[v].z), fmaf(e.w, e.w, s2[v].w));", "")])
    variant("x5", [("*reinterpret_cast<float2*>(stats + (i * HEADS + hh[v]) * 2) = make_float2((beg < end) ? m[v] : 0.f, denom);",
                    "if (denom == 123456.f) *reinterpret_cast<float2*>(stats + (i * HEADS + hh[v]) * 2) = make_float2((beg < end) ? m[v] : 0.f, denom);")])
finally:
    open(src, "w").write(base)
    subprocess.run(["make", "-C", os.path.dirname(src)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print("source restored, library rebuilt")
