"""``launch_group`` host logic that needs no GPU: when a backward-epilogue job may be deferred, the role records handed to
``qot_run_roles`` (``_lib.Role`` mirrors ``qot_role_t`` of include/qot_gnn.h), and that the library refuses bad tables
before launching anything."""
import ctypes

import pytest
import torch

from gnn_qot_estimation_amd import _lib, launch_group as LG


def test_role_struct_matches_the_header_layout():
    # typedef struct qot_role { int32_t kind; int32_t reserved; const void* p[18]; int64_t i[8]; } qot_role_t;
    assert ctypes.sizeof(_lib.Role) == 4 + 4 + 18 * 8 + 8 * 8
    assert _lib.Role.p.offset == 8 and _lib.Role.i.offset == 8 + 18 * 8
    t = torch.zeros(4)
    r = _lib.make_role(_lib.ROLE_SUM_ROWS, (t, None, 12345), (3, 4, 0))
    assert r.kind == _lib.ROLE_SUM_ROWS and r.p[0] == t.data_ptr() and r.p[1] is None and r.p[2] == 12345
    assert list(r.i)[:3] == [3, 4, 0] and list(r.i)[3:] == [0] * 5
    hdr = open(__import__("os").path.join(__import__("os").path.dirname(_lib.__file__), "..", "include", "qot_gnn.h")).read()
    assert "#define QOT_MAX_ROLES %d" % _lib.MAX_ROLES in hdr
    for name, val in (("QOT_ROLE_CSR_BY_GRAPH", 1), ("QOT_ROLE_TABLE_PROJECT_FWD", 2), ("QOT_ROLE_GATHER3", 3),
                      ("QOT_ROLE_SUM_ROWS", 4), ("QOT_ROLE_NNCONV_FINALIZE64", 5), ("QOT_ROLE_TABLE_PROJECT_BWD", 6)):
        assert f"{name} = {val}" in hdr


def test_a_job_is_deferred_only_when_nothing_can_read_its_output_early(monkeypatch):
    """A deferred gradient is filled at the end of the backward pass.  A leaf without ``.grad`` just keeps the tensor; an
    existing ``.grad`` is accumulated into at once, a non-leaf receiver hands the gradient to the next node at once, and
    ``create_graph`` makes autograd clone: all of those must take the immediate launches (the first version of the epilogue
    queue failed ``test_two_rank_hip_step_matches_single_process`` exactly there)."""
    monkeypatch.delenv("QOT_NO_LAUNCH_GROUPS", raising=False)
    leaf = torch.nn.Parameter(torch.zeros(3))
    with torch.no_grad():
        assert LG.can_defer(leaf, None)
        leaf.grad = torch.zeros(3)
        assert not LG.can_defer(leaf)                 # accumulation reads the incoming gradient immediately
        leaf.grad = None
        nonleaf = torch.nn.functional.pad(torch.nn.Parameter(torch.zeros(3)), (0, 1))
    assert not LG.can_defer(nonleaf)                  # F.pad of a parameter (padded widths): PadBackward reads at once
    with torch.enable_grad():
        assert not LG.can_defer(leaf)                 # grad mode on inside backward = create_graph: AccumulateGrad clones
    with torch.no_grad():
        monkeypatch.setenv("QOT_NO_LAUNCH_GROUPS", "1")
        assert not LG.can_defer(leaf) and not LG.enabled()


def test_run_roles_refuses_bad_tables_without_a_gpu():
    lib = _lib.load()
    assert lib.qot_run_roles(None, 0, None) == 0
    arr = (_lib.Role * 1)(_lib.make_role(99, (), ()))
    assert lib.qot_run_roles(ctypes.addressof(arr), 1, None) < 0                  # unknown kind
    arr = (_lib.Role * 1)(_lib.make_role(_lib.ROLE_SUM_ROWS, (None, None), (4, 4, 0)))
    assert lib.qot_run_roles(ctypes.addressof(arr), 1, None) < 0                  # NULL operands
    assert lib.qot_run_roles(ctypes.addressof(arr), _lib.MAX_ROLES + 1, None) < 0
    with pytest.raises(_lib.QotError):
        _lib.check(lib.qot_run_roles(ctypes.addressof(arr), 1, None), "qot_run_roles")


def test_dropping_stale_jobs_clears_the_queue():
    LG._Q.stages[0].append(_lib.make_role(_lib.ROLE_SUM_ROWS, (), ()))
    LG._Q.armed = True
    LG.drop_stale()
    assert not LG._Q.pending() and not LG._Q.armed
