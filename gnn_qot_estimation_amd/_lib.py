"""ctypes binding of ``libqot_gnn.so`` (the C ABI declared in ``include/qot_gnn.h``).

There is no CPU fallback: if the shared library is missing or a call returns non-zero the
caller gets an exception.  Tensors cross the boundary as raw device pointers plus sizes;
the current torch stream is passed so launches are ordered with torch's own work (and are
captured by ``torch.cuda.CUDAGraph``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# QOT_LIB_PATH: a diagnostic build (csrc `make DIAG=1 OUT=...`, tools/ablate_*.py); never set in production
LIB_PATH = os.environ.get("QOT_LIB_PATH") or os.path.join(_HERE, "libqot_gnn.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_f = C.c_float
_u64 = C.c_uint64
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol include/qot_gnn.h declares
ABI_VERSION = 10         # include/qot_gnn.h: QOT_ABI_VERSION

SIGNATURES = {
    "qot_abi_version": (_int, []),
    "qot_error_string": (C.c_char_p, [_int]),
    "qot_csr_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "qot_csr_build": (_int, [_p, _i64, _i64, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "qot_csr_gat_by_graph_supported": (_int, [_i64, _i64]),
    "qot_csr_build_gat_by_graph": (_int, [_p, _i64, _i64, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                          _p]),
    "qot_csr_build_by_graph": (_int, [_p, _i64, _i64, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                      _p, _p, _p, _p, _p, _p, _p]),
    "qot_i32_gather": (_int, [_p, _p, _p, _i64, _p]),
    "qot_i64_to_i32": (_int, [_p, _p, _i64, _p]),
    "qot_batch_ptr": (_int, [_p, _i64, _i64, _p, _p]),
    "qot_embed_fwd": (_int, [_p, _p, _p, _i64, _int, _int, _p]),
    "qot_embed_bwd": (_int, [_p, _p, _p, _i64, _int, _int, _p]),
    "qot_tconv_fwd_scores": (_int, [_p, _p, _p, _int, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int,
                                    _int, _f, _f, _u64, _p, _p]),
    "qot_tconv_rows_supported": (_int, [_int, _int, _int]),
    "qot_tconv_rows_npad": (_int, [_int]),
    "qot_tconv_rows_ld": (_int, [_int, _int, _int]),
    "qot_tconv_bwd_dst_rows": (_int, [_p, _p, _p, _int, _p, _p, _p, _p, _p, _p, _p, _int, _p, _p, _p, _p, _f, _f, _u64, _p,
                                      _int, _i64, _int, _p, _p, _int, _int, _p]),
    "qot_tconv_fwd_rows": (_int, [_p, _p, _p, _int, _p, _int, _p, _p, _p, _p, _p, _p, _p, _int, _i64, _int, _int, _int,
                                  _int, _f, _f, _u64, _p, _p]),
    "qot_tconv_bwd_src_rows": (_int, [_p, _p, _p, _p, _p, _int, _i64, _int, _p, _int, _p]),
    "qot_tconv_fwd": (_int, [_p, _p, _p, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int,
                             _int, _f, _f, _u64, _p, _p]),
    "qot_tconv_fwd_tile": (_int, [_p, _p, _p, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _int, _i64,
                                  _int, _f, _f, _u64, _p, _p]),
    "qot_tconv_bwd_dst_workspace_floats": (_sz, [_i64, _int, _int]),
    "qot_tconv_rows_per_block": (_int, [_int]),
    "qot_tconv_bwd_dst_blocks": (_i64, [_i64, _int, _int, _i64]),
    "qot_tconv_bwd_dst": (_int, [_p, _p, _p, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _p, _p, _p, _p,
                                 _p, _f, _f, _u64, _p, _p, _p, _int, _i64, _p, _i64, _int, _int, _p]),
    "qot_tconv_bwd_src": (_int, [_p, _int, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _int, _int, _i64, _p, _i64, _int,
                                 _p]),
    "qot_tconv_graph_supported": (_int, [_int, _int, _int, _int]),
    "qot_tconv_graph_ldm": (_int, [_int]),
    "qot_tconv_graph_row_floats": (_sz, [_int, _int, _int]),
    "qot_tconv_bwd_graph_blocks": (_int, [_i64]),
    "qot_table_scores": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _int, _int, _int, _p]),
    "qot_tconv_fwd_graph": (_int, [_p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _i64, _int, _int, _int,
                                   _int, _f, _f, _u64, _p, _p]),
    "qot_tconv_bwd_graph": (_int, [_p, _p, _f, _f, _u64, _p, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int,
                                   _i64, _int, _int, _int, _p]),
    "qot_table_project_bwd_scores": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _int, _int, _int, _p]),
    "qot_nnconv_agg": (_int, [_p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _int, _p, _i64, _int, _int, _p]),
    "qot_nnconv_fused": (_int, [_p, _int, _p, _p, _p, _p, _p, _p, _p, _int, _p, _p, _p, _i64, _int, _int,
                                _int, _f, _f, _u64, _p, _p]),
    "qot_nnconv_bwd_finalize": (_int, [_p, _p, _p, _p, _p, _i64, _int, _int, _p]),
    "qot_nnconv_dw_workspace_floats": (_sz, [_i64, _int, _int]),
    "qot_nnconv_dw": (_int, [_p, _int, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _p]),
    "qot_gemm_tn_workspace_floats": (_sz, [_int]),
    "qot_gemm_tn": (_int, [_p, _int, _p, _int, _i64, _int, _p, _p, _p]),
    "qot_nnconv_adjoint_dw_workspace_floats": (_sz, [_int]),
    "qot_nnconv_adjoint_dw": (_int, [_p, _int, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _p, _i64,
                                     _int, _int, _p]),
    "qot_nnconv_gradh_workspace_floats": (_sz, [_int]),
    "qot_nnconv_gradh_fused": (_int, [_p, _int, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int,
                                      _int, _p]),
    "qot_nnconv_bwd_edge": (_int, [_p, _int, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int,
                                   _int, _p]),
    "qot_act_fwd": (_int, [_p, _p, _i64, _f, _f, _u64, _p, _p]),
    "qot_act_bwd": (_int, [_p, _p, _p, _i64, _f, _f, _u64, _p, _p]),
    "qot_pool_fwd": (_int, [_p, _p, _p, _i64, _int, _p]),
    "qot_pool_bwd": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p]),
    "qot_gat_blocks": (_int, [_i64, _int, _int]),
    "qot_gat_bn_partials_floats": (_sz, [_i64, _int, _int]),
    "qot_gat_logits": (_int, [_p, _p, _p, _p, _p, _i64, _int, _int, _p]),
    "qot_gat_fwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _p, _p]),
    "qot_gat_att_grad": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _int, _p]),
    "qot_gat_chunk_rows": (_i64, [_i64, _int, _int]),
    "qot_bn_stats_from_partials": (_int, [_p, _p, _int, _i64, _i64, _int, _f, _f, _p, _p, _p, _p, _p]),
    "qot_gat_thin_supported": (_int, [_int, _int, _int]),
    "qot_gat_fwd_thin": (_int, [_p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _p, _p]),
    "qot_gat_bwd_dst_thin": (_int, [_p, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _p, _p, _p]),
    "qot_gat_bwd_dst": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _p, _p, _p]),
    "qot_gat_bwd_src": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _p, _p, _p, _p]),
    "qot_bn_partials_floats": (_sz, [_i64, _int]),
    "qot_bn_stats": (_int, [_p, _i64, _int, _f, _f, _p, _p, _p, _p, _p, _p]),
    "qot_bn_apply": (_int, [_p, _p, _p, _p, _p, _p, _i64, _int, _int, _p]),
    "qot_bn_bwd_reduce": (_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _p, _p, _p, _p]),
    "qot_bn_bwd_apply": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _int, _p, _p]),
    "qot_bn_apply_rows": (_int, [_p, _p, _i64, _p, _p, _p, _p, _p, _int, _int, _p]),
    "qot_bn_bwd_reduce_rows": (_int, [_p, _p, _i64, _p, _p, _p, _p, _p, _int, _int, _p, _p, _p, _p]),
    "qot_bn_bwd_apply_rows": (_int, [_p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _p, _p]),
    "qot_sgd_momentum": (_int, [_p, _p, _p, _i64, _f, _f, _int, _p]),
    "qot_sgd_momentum_multi": (_int, [_p, _p, _p, _int, _p, _p, _i64, _f, _p, _f, _int, _p]),
    "qot_sgd_momentum_dev": (_int, [_p, _p, _p, _i64, _p, _f, _int, _p]),
    "qot_small_gemm": (_int, [_p, _i64, _i64, _p, _i64, _i64, _p, _p, _int, _int, _int, _int, _int, _p]),
    "qot_colsum_workspace_floats": (_sz, [_int]),
    "qot_colsum": (_int, [_p, _int, _i64, _int, _p, _p, _p]),
    "qot_rowsum_wide_workspace_floats": (_sz, [_i64]),
    "qot_rowsum_wide": (_int, [_p, _i64, _i64, _p, _p, _p]),
    "qot_head_fwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _f, _u64, _p, _p]),
    "qot_head_fwd_loss": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _f, _u64, _p, _p, _f, _p, _p, _p]),
    "qot_head_train": (_int, [_p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _f, _u64, _p, _int,
                              _f, _f, _u64, _p, _p]),
    "qot_head_train_blocks": (_int, [_i64, _int]),
    "qot_head_bwd_workspace_floats": (_sz, [_int, _int]),
    "qot_head_bwd_blocks": (_int, [_i64]),
    "qot_head_bwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _int, _int, _f, _f, _u64, _p,
                            _p, _f, _f, _u64, _p, _p]),
    "qot_act_bwd_colsum": (_int, [_p, _p, _p, _i64, _int, _f, _f, _u64, _p, _p, _p, _p]),
    "qot_smooth_l1_workspace_floats": (_sz, []),
    "qot_smooth_l1": (_int, [_p, _p, _i64, _f, _p, _p, _p, _p]),
    "qot_table_project_fwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _int, _p, _p, _p]),
    "qot_table_project_bwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _int, _int, _p]),
    "qot_table_maps": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "qot_step_advance": (_int, [_p, _p, _p]),
    "qot_gather3": (_int, [_p, _i64, _p, _i64, _p, _p, _p, _i64, _p]),
    "qot_gemm_nt": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _int, _int, _p, _p, _p, _p]),
    "qot_gemm_nt_planes": (_int, [_p, _i64, _p, _i64, _p, _i64, _int, _int, _int, _p]),
    "qot_gemm_tn_splits": (_int, [_int, _int, _i64]),
    "qot_gemm256_takes": (_int, [_i64, _int]),
    "qot_gemm_tn_planes": (_int, [_p, _i64, _p, _i64, _p, _int, _int, _i64, _int, _p, _p, _p]),
    "qot_skinny_linear_fwd": (_int, [_p, _p, _p, _i64, _int, _int, _p]),
    "qot_skinny_linear_fwd_logits": (_int, [_p, _p, _p, _i64, _int, _int, _p, _p, _p, _p, _p]),
    "qot_gemm_nt_logits": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _int, _int, _p, _p, _p, _p, _p, _p, _p, _p]),
    "qot_skinny_linear_dw_blocks": (_int, [_i64]),
    "qot_skinny_linear_dw": (_int, [_p, _p, _p, _i64, _int, _int, _p]),
    "qot_run_roles": (_int, [_p, _int, _p]),
    "qot_rows_gather": (_int, [_p, _p, _p, _i64, _int, _p]),
    "qot_rows_scatter": (_int, [_p, _p, _p, _i64, _int, _p]),
}

MAX_ROLES = 12           # include/qot_gnn.h: QOT_MAX_ROLES
(ROLE_CSR_BY_GRAPH, ROLE_TABLE_PROJECT_FWD, ROLE_GATHER3, ROLE_SUM_ROWS, ROLE_NNCONV_FINALIZE64, ROLE_TABLE_PROJECT_BWD,
 ROLE_TABLE_SCORES, ROLE_TABLE_PROJECT_BWD_SCORES) = range(1, 9)


class Role(C.Structure):
    """``qot_role_t`` (include/qot_gnn.h): one job of a multi-role launch."""
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("p", C.c_void_p * 18), ("i", C.c_int64 * 8)]


def make_role(kind: int, ptrs, ints) -> Role:
    """``ptrs``: tensors / raw pointers / None in the order include/qot_gnn.h lists for ``kind``; ``ints`` likewise."""
    r = Role()
    r.kind = kind
    for k, t in enumerate(ptrs):
        r.p[k] = None if t is None else (t.data_ptr() if isinstance(t, torch.Tensor) else int(t))
    for k, v in enumerate(ints):
        r.i[k] = int(v)
    return r


def run_roles(roles, stream_handle=None):
    """One launch for all of ``roles`` (independent jobs); raises on a non-zero status."""
    if not roles:
        return
    if len(roles) > MAX_ROLES:
        for k in range(0, len(roles), MAX_ROLES):
            run_roles(roles[k:k + MAX_ROLES], stream_handle)
        return
    arr = (Role * len(roles))(*roles)
    code = (_lib or load()).qot_run_roles(C.addressof(arr), len(roles), stream() if stream_handle is None else stream_handle)
    if code != 0:
        check(code, "qot_run_roles")


_lib = None


class QotError(RuntimeError):
    """A libqot_gnn entry point returned a non-zero status."""


def build_library(verbose: bool = False) -> str:
    """Compile ``csrc/*.hip`` for gfx950 into ``libqot_gnn.so`` (in-tree)."""
    cmd = ["make", "-C", CSRC_DIR, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libqot_gnn.so failed:\n" + res.stdout[-4000:] + res.stderr[-8000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def load():
    """Load the shared library and bind every declared symbol; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QotError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gnn_qot_estimation_amd/csrc` -- there is no CPU fallback for the HIP path"
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.qot_abi_version() != ABI_VERSION:
        raise QotError(f"libqot_gnn ABI version {lib.qot_abi_version()} != {ABI_VERSION} expected by this package: "
                       "rebuild with `make -C gnn_qot_estimation_amd/csrc`")
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Raw handle of the current stream.  ``torch.cuda.current_stream()`` builds a Python Stream object
    (~10 us per call, a fifth of an eager step at the reference's own batch size: tools/profile_host.py);
    the raw accessor is the same lookup without the object."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def check(code: int, what: str):
    if code != 0:
        msg = load().qot_error_string(code).decode()
        raise QotError(f"{what} failed with status {code}: {msg}")


def call(name: str, *args):
    """Invoke ``name`` with the current stream appended; raise on a non-zero status.

    Arguments may be tensors: they are passed as their device pointers and stay referenced by ``args`` until the
    launch has been enqueued, so ``call(name, t.contiguous(), ...)`` is safe where ``call(name, ptr(t.contiguous()),
    ...)`` is NOT (``ptr`` returns a bare int: the temporary dies at once and the caching allocator may hand its
    block to the next temporary of the same statement -- the round-2 GPU fault)."""
    if any(isinstance(a, torch.Tensor) for a in args):
        raw = tuple(a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args)
    else:
        raw = args
    code = getattr(_lib or load(), name)(*raw, stream())
    if code != 0:
        check(code, name)
