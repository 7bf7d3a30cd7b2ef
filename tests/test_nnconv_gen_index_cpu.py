"""Host-side operand packing of the width-generic fused NNConv kernels (``functional.nnconv_gen_indices``) against
the layouts documented in csrc/nnconv_gen.hip: unpacking the fragment-ordered buffers must give back Wcat, its
per-block transposes and Wk^T exactly (integer-valued data: any misplaced element shows)."""
import pytest
import torch

from gnn_qot_estimation_amd.functional import nnconv_gen_indices, nnconv_wcat, nnconv_wcat_t


@pytest.mark.parametrize("h", [16, 32, 128, 256])
@pytest.mark.parametrize("d", [4, 2])
def test_generic_fragment_orders_unpack_to_wcat(h, d):
    k = 2 * d
    w2 = torch.arange(h * h * k, dtype=torch.float32).view(h * h, k) + 1
    b2 = -(torch.arange(h * h, dtype=torch.float32) + 1)
    wroot = (torch.arange(h * h, dtype=torch.float32).view(h, h) + 1) * 0.5
    flat = torch.cat([w2.reshape(-1), b2, wroot.reshape(-1), torch.zeros(1)])
    idx, n_f, n_a, n_g = nnconv_gen_indices(h, k, "cpu")
    idx = idx.long()
    packed = torch.where(idx < 0, torch.zeros(()), flat[idx.clamp(min=0)])
    wcat, wcat_t = nnconv_wcat(w2, b2, wroot, h, h, k), nnconv_wcat_t(w2, b2, wroot, h, h, k)
    cw, ncb = min(h, 64), (h + 31) // 32
    n_pass, gall = h // cw, (k + 2) * cw // 8
    for name, buf, ref in (("fwd", packed[:n_f], wcat), ("adj", packed[n_f:n_f + n_a], wcat_t)):
        v = buf.view(n_pass, ncb, gall, 64, 4)
        got = torch.zeros((k + 2) * h, ncb * 32)
        P, CB, G, L, R = torch.meshgrid(torch.arange(n_pass), torch.arange(ncb), torch.arange(gall), torch.arange(64),
                                        torch.arange(4), indexing="ij")
        kl = 8 * G + 2 * R + (L >> 5)
        row = (kl // cw) * h + P * cw + kl % cw
        col = CB * 32 + (L & 31)
        got[row.reshape(-1), col.reshape(-1)] = v.reshape(-1)
        assert torch.equal(got[:, :h], ref), name
        assert not got[:, h:].any()                      # padding columns (H = 16) are zeros
    cwg = min(h, 32)
    npg, nbg, gh = h // cwg, k * cwg // 32, h // 8
    v = packed[n_f + n_a:].view(npg, nbg, gh, 64, 4)
    P, NB, GQ, L, R = torch.meshgrid(torch.arange(npg), torch.arange(nbg), torch.arange(gh), torch.arange(64),
                                     torch.arange(4), indexing="ij")
    o = 8 * GQ + 2 * R + (L >> 5)
    n = NB * 32 + (L & 31)
    kq, a = n // cwg, P * cwg + n % cwg
    # GA[i, k, a] = sum_o g_i[o] W2[a*h + o, k]
    assert torch.equal(v, w2[(a * h + o).reshape(-1), kq.reshape(-1)].view_as(v))
    assert n_g == k * h * h
