"""CPU oracle for the QoT-GNN message-passing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gnn_qot_estimation_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.

What it restates
----------------
The reference's arithmetic lives in a third-party dependency that is absent from
``/root/reference`` and from this image: ``torch_geometric`` (version unpinned by the
reference -- no requirements file; checkpoint key names are consistent with PyG >= 2.5,
most likely 2.6.x).  Call sites: ``topological_training/models.py:3,15-17,25-30,61`` and
``lightpath_training/models.py:3,13-14``.  The published operator definitions
 are restated here twice:

* ``oracle.sparse``  -- fp32, edge-list form, built from exactly the torch primitives
  PyG lowers to without torch_scatter (``index_select``, ``scatter_add_``,
  ``scatter_reduce_('amax')``, a materialised ``[E, H*H]`` NNConv weight tensor).
  This is also the timed CPU baseline (``cpu_baseline.kind == "port"``).
* ``oracle.dense64`` -- fp64, an independent per-graph dense masked restatement that
  shares no code with ``oracle.sparse`` (loops over graphs, adjacency-count matrices).

PARITY PIN STATUS: **parity unpinned by the reference's own tests** -- the reference has
no tests, golden vectors or fixtures for this path and PyG cannot be
imported here (``ModuleNotFoundError``, an ordinary Python error; nothing was
permission-denied).  What pins the oracle instead: (1) hand-derived known-answer tests
(``tests/test_oracle_kat.py``), (2) agreement of the two restatements to <=1e-6 rel,
(3) ``torch.autograd.gradcheck`` on the fp64 path, (4) strict ``load_state_dict`` of the
three shipped checkpoints (App. A key/shape contract).
"""
