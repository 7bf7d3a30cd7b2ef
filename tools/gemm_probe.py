"""Time the dense GEMM shapes of the cfg2 step under rocBLAS vs hipBLASLt (fp32)."""
import torch
dev = torch.device("cuda:0")
N, H, K = 102400, 64, 8
shapes = {
    "qkvs fwd   [N,64]x[64,256]": (lambda: (torch.randn(N, 64, device=dev), torch.randn(64, 256, device=dev))),
    "qkvs dX    [N,256]x[256,64]": (lambda: (torch.randn(N, 256, device=dev), torch.randn(256, 64, device=dev))),
    "qkvs dW    [256,N]x[N,64]": (lambda: (torch.randn(N, 256, device=dev).t(), torch.randn(N, 64, device=dev))),
    "nn A@Wcat  [N,640]x[640,64]": (lambda: (torch.randn(N, 640, device=dev), torch.randn(640, 64, device=dev))),
    "nn dWcat   [640,N]x[N,64]": (lambda: (torch.randn(N, 640, device=dev).t(), torch.randn(N, 64, device=dev))),
    "nn GA      [N,64]x[64,512]": (lambda: (torch.randn(N, 64, device=dev), torch.randn(64, 512, device=dev))),
}
def t(fn, it=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
for lib in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as ex:
        print(lib, "unavailable", ex); continue
    for name, mk in shapes.items():
        a, b = mk()
        us = t(lambda: a @ b)
        fl = 2.0 * a.shape[0] * a.shape[1] * b.shape[1]
        print(f"{lib:9s} {name:32s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
