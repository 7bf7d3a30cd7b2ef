"""Shared helpers for parity tests: relative error, weight sharing oracle <-> HIP modules."""
import torch


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max(|b|_inf, tiny): the <=1e-4 'rel fp32' bar of BASELINE.json's north_star."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if a.numel() == 0 and b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


TOL = 1e-4  # BASELINE.json north_star: outputs match the reference forward to <=1e-4 rel fp32
