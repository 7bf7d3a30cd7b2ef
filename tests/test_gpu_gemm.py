"""csrc/gemm.hip: the dense fp32 projections of LightpathGNN (lightpath_training/models.py:13,30 and their autograd)
against fp64 torch products.  Tolerance: fp32 accumulation over K terms, error <= 1e-5 of the result scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


# the last three sizes run on the 256 x 256 tiles of gemm256.hip (>= one tile per CU): ragged M, ragged N, two stages
@pytest.mark.parametrize("M,N,K", [(1000, 512, 512), (128, 128, 32), (77, 512, 64), (4099, 128, 512), (1, 32, 32),
                                   (33003, 512, 64), (66010, 260, 96), (40000, 512, 32)])
@pytest.mark.parametrize("affine", [False, True])
def test_gemm_nt_matches_fp64(cuda_device, M, N, K, affine):
    from gnn_qot_estimation_amd import _lib
    torch.manual_seed(0)
    dev = cuda_device
    assert _lib.load().qot_gemm256_takes(M, N) == (1 if M > 30000 else 0)
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    bias = torch.randn(N, device=dev)
    scale, shift = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)) if affine else (None, None)
    C = torch.full((M, N), float("nan"), device=dev)
    _lib.call("qot_gemm_nt", A, K, B, K, C, N, M, N, K, scale, shift, bias)
    Ad = torch.relu(A.double() * scale.double() + shift.double()) if affine else A.double()
    ref = Ad @ B.double().t() + bias.double()
    assert _rel(C, ref) <= 1e-5
    # strided operands (column slices of wider matrices), no bias
    Aw, Bw = torch.randn(M, K + 64, device=dev), torch.randn(N, K + 32, device=dev)
    Cw = torch.full((M, N + 8), float("nan"), device=dev)
    _lib.call("qot_gemm_nt", Aw, K + 64, Bw, K + 32, Cw, N + 8, M, N, K, None, None, None)
    assert _rel(Cw[:, :N], Aw[:, :K].double() @ Bw[:, :K].double().t()) <= 1e-5
    assert bool(torch.isnan(Cw[:, N:]).all())


# the last two: ragged chunk ends, ragged M / N at a long K
# (K a multiple of 32: the branch-free, pinned kernel gemm_tn_full_kernel; else the general one)
@pytest.mark.parametrize("M,N,K", [(512, 512, 5000), (128, 512, 33), (512, 128, 100000), (4, 8, 7), (512, 512, 70001),
                                   (260, 388, 66000), (260, 388, 64000), (512, 512, 9600), (128, 128, 32)])
@pytest.mark.parametrize("affine", [False, True])
def test_gemm_tn_planes_sum_to_the_product(cuda_device, M, N, K, affine):
    from gnn_qot_estimation_amd import _lib
    torch.manual_seed(1)
    dev = cuda_device
    A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    scale, shift = (torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)) if affine else (None, None)
    splits = _lib.load().qot_gemm_tn_splits(M, N, K)
    assert splits >= 1
    part = torch.full((splits, M * N), float("nan"), device=dev)
    _lib.call("qot_gemm_tn_planes", A, M, B, N, part, M, N, K, splits, scale, shift)
    out = torch.empty(M * N, device=dev)
    _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (part, out), (splits, M * N, 0))])
    Bd = torch.relu(B.double() * scale.double() + shift.double()) if affine else B.double()
    ref = A.double().t() @ Bd
    assert _rel(out.view(M, N), ref) <= 2e-5
    part2 = torch.empty_like(part)
    _lib.call("qot_gemm_tn_planes", A, M, B, N, part2, M, N, K, splits, scale, shift)
    assert torch.equal(part, part2)                          # fixed order: bitwise reproducible


@pytest.mark.parametrize("training", [True, False])
def test_bn_relu_folded_into_the_next_projection_equals_the_materialised_path(cuda_device, training):
    """``QF.BnLinearFn`` (opt-in ``_qot_fuse_bn_projection``): ``relu(norm_l(x)) @ W_{l+1}^T`` with the normalised
    activations never materialised -- BatchNorm + ReLU applied in the operand load of ``qot_gemm_nt``, recomputed in
    ``qot_gemm_tn_planes`` for the weight gradient -- against the default path (BnFn, library projections) and the
    oracle: outputs, every gradient, BatchNorm running statistics (3 layers, C = 32: inner width 128)."""
    import copy
    import gnn_qot_estimation_amd as q
    from gnn_qot_estimation_amd import synthetic as S
    from oracle import sparse as O
    from helpers import TOL, rel_err
    torch.manual_seed(0)
    ref = O.LightpathGNN(5, 32, 3, 1, dropout_p=0.0, num_layers=3)
    a = q.LightpathGNN(5, 32, 3, 1, dropout_p=0.0, num_layers=3)
    a.load_state_dict(ref.state_dict(), strict=True)
    a.to(cuda_device)
    b = copy.deepcopy(a)
    a._qot_fuse_bn_projection, b._qot_fuse_bn_projection = True, False
    batch = S.lightpath_batch(40)
    dbatch = batch.to(cuda_device)
    for m in (ref, a, b):
        m.train(training)
    outs = []
    for m, d in ((ref, batch), (a, dbatch), (b, dbatch)):
        o, lb = m(d)
        if training:
            torch.nn.functional.smooth_l1_loss(o, d.y[lb]).backward()
        outs.append(o.detach().cpu())
    assert rel_err(outs[1], outs[0]) <= TOL and rel_err(outs[2], outs[0]) <= TOL
    if training:
        rp = dict(ref.named_parameters())
        gmax = max(float(p.grad.abs().max()) for p in rp.values())
        for name, p in a.named_parameters():
            floor = gmax if name.endswith("conv1.bias") or name.endswith("conv2.bias") or name.endswith("conv3.bias") else 1e-3 * gmax
            e = float((p.grad.cpu() - rp[name].grad).abs().max() / max(float(rp[name].grad.abs().max()), floor))
            assert e <= TOL, (name, e)
        for k, v in ref.state_dict().items():
            if "running" in k:
                assert rel_err(a.state_dict()[k].cpu(), v) <= TOL, k


@pytest.mark.parametrize("N,F,C", [(1000, 5, 512), (33, 5, 128), (4097, 8, 32), (7, 1, 1024), (100000, 5, 512)])
def test_skinny_first_layer_projection_and_weight_gradient(cuda_device, N, F, C):
    """``QF.SkinnyLinearFn`` (GATConv's first-layer ``lin``: 5 raw node features in, lightpath_training/models.py:13) against
    fp64: forward and weight gradient; bitwise run to run."""
    from gnn_qot_estimation_amd import functional as QF
    torch.manual_seed(0)
    x = torch.randn(N, F, device=cuda_device)
    w = torch.randn(C, F, device=cuda_device, requires_grad=True)
    g = torch.randn(N, C, device=cuda_device)
    out = QF.SkinnyLinearFn.apply(x, w)
    out.backward(g)
    ref = x.double() @ w.detach().double().t()
    assert _rel(out.detach(), ref) <= 1e-6
    assert _rel(w.grad, g.double().t() @ x.double()) <= 2e-5
    g1 = w.grad.clone()
    w.grad = None
    QF.SkinnyLinearFn.apply(x, w).backward(g)
    assert torch.equal(g1, w.grad)


@pytest.mark.parametrize("M,heads,K", [(1000, 4, 512), (77, 1, 64), (4099, 2, 128), (33003, 4, 64), (70001, 3, 32)])
@pytest.mark.parametrize("affine", [False, True])
def test_gemm_nt_logits_epilogue_matches_fp64(cuda_device, M, heads, K, affine):
    """qot_gemm_nt_logits: the product and GATConv's attention logits a[m, h] = <out[m, h, :], att[h, :]> (128 channels per
    head, PyG GATConv alpha_src / alpha_dst) from the same launch."""
    from gnn_qot_estimation_amd import _lib
    torch.manual_seed(2)
    dev = cuda_device
    N = heads * 128
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5
    scale, shift = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)) if affine else (None, None)
    att_s, att_d = torch.randn(N, device=dev), torch.randn(N, device=dev)
    C = torch.full((M, N), float("nan"), device=dev)
    a_s, a_d = torch.full((M, heads), float("nan"), device=dev), torch.full((M, heads), float("nan"), device=dev)
    _lib.call("qot_gemm_nt_logits", A, K, B, K, C, N, M, N, K, scale, shift, None, att_s, att_d, a_s, a_d)
    Ad = torch.relu(A.double() * scale.double() + shift.double()) if affine else A.double()
    ref = Ad @ B.double().t()
    assert _rel(C, ref) <= 1e-5
    assert _rel(a_s, (ref.view(M, heads, 128) * att_s.double().view(heads, 128)).sum(-1)) <= 1e-5
    assert _rel(a_d, (ref.view(M, heads, 128) * att_d.double().view(heads, 128)).sum(-1)) <= 1e-5
    # a width that is not one tile per head is refused, not mis-computed
    with pytest.raises(_lib.QotError):
        _lib.call("qot_gemm_nt_logits", A, K, B, K, C, N - 64, M, N - 64, K, scale, shift, None, att_s, att_d, a_s, a_d)


@pytest.mark.parametrize("N,F,C", [(1001, 5, 512), (64, 8, 128), (300, 3, 256)])
def test_skinny_projection_logits_match_fp64(cuda_device, N, F, C):
    from gnn_qot_estimation_amd import _lib
    torch.manual_seed(3)
    dev = cuda_device
    heads = C // 128
    x, w = torch.randn(N, F, device=dev), torch.randn(C, F, device=dev)
    att_s, att_d = torch.randn(C, device=dev), torch.randn(C, device=dev)
    out = torch.full((N, C), float("nan"), device=dev)
    a_s, a_d = torch.full((N, heads), float("nan"), device=dev), torch.full((N, heads), float("nan"), device=dev)
    _lib.call("qot_skinny_linear_fwd_logits", x, w, out, N, F, C, att_s, att_d, a_s, a_d)
    ref = x.double() @ w.double().t()
    assert _rel(out, ref) <= 1e-5
    assert _rel(a_s, (ref.view(N, heads, 128) * att_s.double().view(heads, 128)).sum(-1)) <= 1e-5
    assert _rel(a_d, (ref.view(N, heads, 128) * att_d.double().view(heads, 128)).sum(-1)) <= 1e-5


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("N,C,n", [(5000, 512, 613), (300, 128, 300), (77, 32, 1)])
def test_batchnorm_rows_form_equals_dense_batchnorm_then_gather(cuda_device, training, N, C, n):
    """QF.BnRowsFn (only the consumed rows of relu(BatchNorm(x)) are formed; lightpath_training/models.py:31-32, 35-40)
    against BnFn + RowsGatherFn: outputs and running statistics bit for bit, parameter gradients to fp32 rounding (same
    terms, other order), grad_x bit for bit once both paths are given the same column sums."""
    from gnn_qot_estimation_amd import functional as QF, _lib
    torch.manual_seed(3)
    dev = cuda_device
    x0 = torch.randn(N, C, device=dev) * 2 + 0.5
    idx = torch.randperm(N, device=dev)[:n].sort().values.to(torch.int32)
    g = torch.randn(n, C, device=dev)
    w0, b0 = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    res = []
    for rows in (False, True):
        x = x0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        if rows:
            y = QF.BnRowsFn.apply(x, w, b, rm, rv, training, 0.1, 1e-5, True, False, None, idx)
        else:
            y = QF.RowsGatherFn.apply(QF.BnFn.apply(x, w, b, rm, rv, training, 0.1, 1e-5, True, False, None), idx)
        y.backward(g)
        res.append((y.detach(), rm, rv, x.grad, w.grad, b.grad))
    a, r = res
    assert torch.equal(a[0], r[0]) and torch.equal(a[1], r[1]) and torch.equal(a[2], r[2])
    for k in (4, 5):
        assert float((a[k] - r[k]).abs().max()) <= 2e-5 * float(a[k].abs().max().clamp_min(1.0))
    assert float((a[3] - r[3]).abs().max()) <= 1e-5 * float(a[3].abs().max().clamp_min(1e-3))
    if training:
        # same column sums in: the two-pass rows kernel reproduces the dense kernel's bits
        P = lambda t: t
        x = x0
        mean, var = x.mean(0), x.var(0, unbiased=False)
        rstd = (var + 1e-5).rsqrt()
        w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        gw, gb = torch.randn(C, device=dev), torch.randn(C, device=dev)
        gfull = torch.zeros(N, C, device=dev)
        gfull[idx.long()] = g
        dense, rowsf = torch.empty(N, C, device=dev), torch.full((N, C), float("nan"), device=dev)
        _lib.call("qot_bn_bwd_apply", gfull, None, x, mean, rstd, w, gw, gb, dense, N, C, 1, 1, b)
        _lib.call("qot_bn_bwd_apply_rows", g, idx, n, x, mean, rstd, w, gw, gb, rowsf, N, C, 1, b)
        assert torch.equal(dense, rowsf)


@pytest.mark.parametrize("m,k,n", [(65536, 128, 1), (5000, 64, 3), (4096, 512, 8)])
def test_small_linear_thin_output_weight_gradient_matches_fp64(cuda_device, m, k, n):
    """SmallLinearFn for a handful of outputs over many rows (LightpathGNN's last Linear, models.py:17-22): the weight
    gradient goes through the skinny row-parallel kernel with the operands' roles swapped."""
    from gnn_qot_estimation_amd import functional as QF
    torch.manual_seed(4)
    dev = cuda_device
    x = torch.randn(m, k, device=dev, requires_grad=True)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).requires_grad_(True)
    b = torch.randn(n, device=dev, requires_grad=True)
    g = torch.randn(m, n, device=dev)
    y = QF.SmallLinearFn.apply(x, w, b)
    y.backward(g)
    xd, wd, gd = x.detach().double(), w.detach().double(), g.double()
    assert _rel(y.detach(), xd @ wd.t() + b.detach().double()) <= 1e-5
    assert _rel(w.grad, gd.t() @ xd) <= 2e-5
    assert _rel(x.grad, gd @ wd) <= 1e-5
    assert _rel(b.grad, gd.sum(0)) <= 2e-5


@pytest.mark.parametrize("M,N,K,ks", [(1000, 128, 1056, 3), (1000, 256, 1024, 8), (77, 64, 256, 4), (4099, 128, 64, 2)])
def test_gemm_nt_planes_sum_to_the_product(cuda_device, M, N, K, ks):
    """qot_gemm_nt_planes: few output tiles, long inner dimension -- plane s is the product over the s-th slice of K; the planes
    summed in order give A B^T (grad T_q of the row form, the table projection's grad_table)."""
    from gnn_qot_estimation_amd import _lib
    torch.manual_seed(6)
    dev = cuda_device
    A, B = torch.randn(M, K + 32, device=dev), torch.randn(N, K, device=dev)       # A: a column slice of a wider matrix
    planes = torch.full((ks, M * N), float("nan"), device=dev)
    _lib.call("qot_gemm_nt_planes", A, K + 32, B, K, planes, M, N, K, ks)
    out = torch.empty(M * N, device=dev)
    _lib.run_roles([_lib.make_role(_lib.ROLE_SUM_ROWS, (planes, out), (ks, M * N, 0))])
    ref = A[:, :K].double() @ B.double().t()
    assert _rel(out.view(M, N), ref) <= 1e-5
    per = K // ks
    assert _rel(planes[1].view(M, N), A[:, per:2 * per].double() @ B[:, per:2 * per].double().t()) <= 1e-5
    with pytest.raises(_lib.QotError):                      # a slice that is not a whole number of 32-deep stages
        _lib.call("qot_gemm_nt_planes", A, K + 32, B, K, planes, M, N, K, ks + 5)
