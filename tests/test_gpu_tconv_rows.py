"""TransformerConv in table mode for large tables (cfg4 / cfg5: V = 1000): logits from the score matrix T_q T_k^T in the
forward (qot_tconv_fwd_scores) and the row form of the backward (csrc/tconv_rows.hip: grad M rows, grad T_q / grad T_k from
two small products) against the per-destination kernels of csrc/tconv.hip on the same batch -- which the oracle tests pin
(test_gpu_configs.py, test_gpu_parity.py).  fp32 sums in another order: 2e-4 of the tensor's scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-4):
    return float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("n,e,H,B,p,D", [(1000, 4000, 128, 8, 0.0, 4), (1000, 4000, 256, 5, 0.0, 4), (600, 1800, 64, 9, 0.5, 4),
                                         (500, 1500, 128, 9, 0.0, 6), (300, 900, 64, 13, 0.0, 1)])
def test_scores_forward_and_row_form_backward_equal_the_per_destination_kernels(cuda_device, monkeypatch, n, e, H, B, p, D):
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import _lib, synthetic as S
    dev = cuda_device
    batch = S.topological_batch(4, B, n=n, e=e, edge_dim=D).to(dev)
    torch.manual_seed(0)
    model = q.TopologicalGNN(n, H, 3, D, dropout_p=p).to(dev).train()
    res = {}
    for mode in ("rows", "plain"):
        calls = []
        real = _lib.call
        monkeypatch.setattr(_lib, "call", lambda name, *a, _c=calls: (_c.append(name), real(name, *a))[1])
        if mode == "plain":
            monkeypatch.setenv("QOT_NO_TCONV_SCORES", "1")
        else:
            monkeypatch.delenv("QOT_NO_TCONV_SCORES", raising=False)
        batch._qot_cache = {}
        model.zero_grad(set_to_none=True)
        q.functional.reset_dropout_state(dev) if hasattr(q.functional, "reset_dropout_state") else None
        out = model(batch)
        out.square().sum().backward()
        monkeypatch.setattr(_lib, "call", real)
        res[mode] = (out.detach().clone(), {k: v.grad.clone() for k, v in model.named_parameters()}, calls)
    assert "qot_tconv_fwd_rows" in res["rows"][2] and "qot_tconv_bwd_dst_rows" in res["rows"][2]
    assert "qot_tconv_fwd_rows" not in res["plain"][2] and "qot_tconv_bwd_dst" in res["plain"][2]
    if p == 0.0:                       # (with dropout the two runs draw different masks: the step counter advances)
        assert _close(res["rows"][0], res["plain"][0])
        for k, g in res["rows"][1].items():
            ref = res["plain"][1][k]
            if k.endswith("lin_key.bias"):
                # analytically zero (a constant added to every key of a destination leaves its softmax alone): both paths hold
                # rounding noise there -- compared on the scale of the key weight's gradient
                scale = float(res["plain"][1][k.replace("bias", "weight")].abs().max())
                assert float((g - ref).abs().max()) <= 2e-4 * scale, k
            else:
                assert _close(g, ref), k


def test_row_form_backward_is_bitwise_reproducible(cuda_device):
    """The lane groups of a workgroup add into one LDS image of a grad M row in no fixed order -- as 64-bit fixed point, so
    two runs of the same step give the same bits."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    dev = cuda_device
    batch = S.topological_batch(5, 6, n=1000).to(dev)              # power-law degrees: hubs share rows across graphs
    torch.manual_seed(1)
    model = q.TopologicalGNN(1000, 128, 3, 4, dropout_p=0.0).to(dev).train()
    grads = []
    for _ in range(2):
        batch._qot_cache = {}
        model.zero_grad(set_to_none=True)
        model(batch).square().sum().backward()
        grads.append([p_.grad.clone() for p_ in model.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_row_form_forward_is_bit_equal_to_the_per_destination_scores_kernel(cuda_device, monkeypatch):
    """qot_tconv_fwd_rows stages the score row, q_r W_e and the skip row once per workgroup; same arithmetic in the same order
    as tconv_fwd_kernel<., ., MT>: identical outputs (dropout on: same counter, same mask)."""
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    dev = cuda_device
    batch = S.topological_batch(5, 7, n=1000).to(dev)
    torch.manual_seed(2)
    model = q.TopologicalGNN(1000, 128, 3, 4, dropout_p=0.0).to(dev).eval()
    outs = []
    for off in ("", "1"):
        monkeypatch.setenv("QOT_NO_TCONV_FWD_ROWS", off) if off else monkeypatch.delenv("QOT_NO_TCONV_FWD_ROWS", raising=False)
        batch._qot_cache = {}
        with torch.no_grad():
            outs.append(model(batch).clone())
    assert torch.equal(outs[0], outs[1])
