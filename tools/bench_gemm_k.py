"""qot_gemm_nt against the library over the inner dimension K at a fixed flop count (N = 512): separates the per-tile costs
(prologue, epilogue: once per 128 x 128 output tile whatever K) from the per-stage costs of the main loop."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_qot_estimation_amd import _lib


def timeit(fn, it=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it


dev = torch.device("cuda")
N = 512
for K in (128, 256, 512, 1024, 2048, 4096):
    M = 707008 * 512 // K // 128 * 128
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    z = torch.empty(M, N, device=dev)
    s, t = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    fl = 2.0 * M * N * K
    lib = timeit(lambda: torch.mm(x, w.t(), out=z))
    ours = timeit(lambda: _lib.call("qot_gemm_nt", x, K, w, K, z, N, M, N, K, None, None, None))
    aff = timeit(lambda: _lib.call("qot_gemm_nt", x, K, w, K, z, N, M, N, K, s, t, None))
    print(json.dumps({"K": K, "M": M, "lib_tf": round(fl / lib / 1e9, 1), "nt_tf": round(fl / ours / 1e9, 1),
                      "nt_affine_tf": round(fl / aff / 1e9, 1)}))
    del x, z

lib = _lib.load()
if hasattr(lib, "qot_debug_gemm_variant"):
    import ctypes
    lib.qot_debug_gemm_variant.argtypes = [ctypes.c_int]
    for K in (512, 4096):
        M = 707008 * 512 // K // 128 * 128
        x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        z = torch.empty(M, N, device=dev)
        s, t = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
        fl = 2.0 * M * N * K
        abl = {}
        for name, v in (("full", 0), ("no global loads", 1), ("no LDS stores", 2), ("no loads, no stores", 3), ("no barrier", 4),
                        ("no loads/stores/barrier", 7), ("no C stores", 8), ("no fragment reads", 16), ("MFMAs only", 31),
                        ("A from one row (L2)", -1), ("C to one row", -2)):
            lib.qot_debug_gemm_variant(max(v, 0))
            if hasattr(lib, "qot_debug_gemm256_variant"):
                lib.qot_debug_gemm256_variant(max(v, 0))
            lda = 0 if v == -1 else K
            ldc = 0 if v == -2 else N
            ms = timeit(lambda: _lib.call("qot_gemm_nt", x, lda, w, K, z, ldc, M, N, K, None, None, None))
            ms2 = timeit(lambda: _lib.call("qot_gemm_nt", x, lda, w, K, z, ldc, M, N, K, s, t, None))
            abl[name] = [round(fl / ms / 1e9, 1), round(fl / ms2 / 1e9, 1)]
        lib.qot_debug_gemm_variant(0)
        if hasattr(lib, "qot_debug_gemm256_variant"):
            lib.qot_debug_gemm256_variant(0)
        print(json.dumps({"K": K, "tile": 256 if lib.qot_gemm256_takes(M, N) else 128, "ablation TF [plain, affine]": abl}))
        del x, z
