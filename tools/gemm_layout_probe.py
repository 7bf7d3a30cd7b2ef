import torch, time
dev = torch.device("cuda:0")
N = 720000
x = torch.randn(N, 512, device=dev); W = torch.randn(512, 512, device=dev); g = torch.randn(N, 512, device=dev)
Wt = W.t().contiguous()
def t(f, it=5):
    f(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(it): f()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / it
fl = 2.0 * N * 512 * 512
for name, f in (("x @ W.t() (NT)", lambda: x @ W.t()), ("x @ Wt contiguous (NN)", lambda: x @ Wt), ("g @ W (NN)", lambda: g @ W),
                ("g.t() @ x (TN)", lambda: g.t() @ x), ("x.t() @ g (TN)", lambda: x.t() @ g),
                ("x5 @ W5.t()", None)):
    if f is None: continue
    ms = t(f); print(f"{name:28s} {ms:7.3f} ms  {fl/ms/1e9:6.1f} TFLOP/s")
x5 = torch.randn(N, 5, device=dev); W5 = torch.randn(512, 5, device=dev)
ms = t(lambda: x5 @ W5.t()); print(f"x5 @ W5.t() [N,5]x[5,512]    {ms:7.3f} ms")
ms = t(lambda: g.t() @ x5); print(f"g.t() @ x5 [512,N]x[N,5]     {ms:7.3f} ms")
wa = torch.randn(8, 512, device=dev)
ms = t(lambda: x @ wa.t()); print(f"x @ w_att.t() [N,512]x[512,8] {ms:7.3f} ms")
ga = torch.randn(N, 8, device=dev)
ms = t(lambda: ga @ wa); print(f"ga @ w_att [N,8]x[8,512]      {ms:7.3f} ms")
ms = t(lambda: ga.t() @ x); print(f"ga.t() @ x [8,N]x[N,512]      {ms:7.3f} ms")
