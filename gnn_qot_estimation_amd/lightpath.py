"""``LightpathGNN`` on the HIP message-passing engine.

Same constructor, ``forward(data) -> (out, lut_batch)`` contract, ``ValueError`` behaviour
and ``state_dict`` keys as ``lightpath_training/models.py:7-45``.

``num_layers`` is a build extension (default 1 = the reference): extra
``GATConv(4C, C, heads=4) + BatchNorm + ReLU`` blocks named ``conv2/norm2``, ... for the
3-layer benchmark configuration.
"""
from __future__ import annotations

import os

from torch import nn

from . import functional as QF
from .graph import _cache, check_index_status, graph_index_for, to_i32
from .nn import BatchNorm, GATConv


class LightpathGNN(nn.Module):
    def __init__(self, in_channels, hidden_channels, output_dim, is_lut_index, dropout_p=0.5, num_layers=1):
        super().__init__()
        if num_layers < 1:
            raise ValueError("num_layers >= 1")
        width = hidden_channels * 4
        for layer in range(1, num_layers + 1):
            setattr(self, f"conv{layer}", GATConv(in_channels if layer == 1 else width, hidden_channels,
                                                  heads=4, concat=True))
            setattr(self, f"norm{layer}", BatchNorm(width))
        self.mlp = nn.Sequential(
            nn.Linear(width, hidden_channels),
            nn.LeakyReLU(),
            nn.Dropout(p=dropout_p),
            nn.Linear(hidden_channels, output_dim),
        )
        self.is_lut_index = is_lut_index
        self.num_layers = num_layers
        # data-parallel shards (harness.run_epoch): the LUT-less test is a property of the GLOBAL batch, so a rank
        # whose shard holds no LUT node returns zero rows instead of raising (plain attribute, not in state_dict)
        self.allow_empty_lut = False
        # per-head widths the GAT kernels are not instantiated for run zero-padded (gnn_qot_estimation_amd/padded.py)
        from .padded import GAT_WIDTHS, padded_width
        self._qot_cp = None if hidden_channels in GAT_WIDTHS else padded_width(hidden_channels, GAT_WIDTHS)
        self._qot_shadow = None

    def _lut_rows(self, data, check: bool = True):
        """Indices of LUT nodes: ``data.x[:, is_lut_index] == 1.0`` on the RAW input
        (models.py:35).  The ``ValueError`` and the data-dependent output length are part
        of the reference contract, so one host sync per distinct batch is unavoidable; the
        result is cached on the batch object."""
        x0 = data.x
        tag = (x0.data_ptr(), x0._version, tuple(x0.shape), int(self.is_lut_index))
        c = _cache(data)
        if c is not None and "lut" in c and c["lut"][0] == tag:
            idx = c["lut"][1]
        else:
            idx = (x0[:, self.is_lut_index] == 1.0).nonzero().squeeze(1)
            if c is not None:
                c["lut"] = (tag, idx)
        if check and idx.numel() == 0 and not self.allow_empty_lut:
            raise ValueError("No LUT node found in the batch.")
        return idx

    def _forward_padded(self, data):
        import torch
        from . import padded
        c, cp = self.conv1.out_channels, self._qot_cp
        dev = self.conv1.bias.device
        if self._qot_shadow is None or self._qot_shadow[0].conv1.bias.device != dev:
            shadow = LightpathGNN(self.conv1.in_channels, cp, self.mlp[3].out_features, self.is_lut_index,
                                  dropout_p=self.mlp[2].p, num_layers=self.num_layers).to(dev)
            for p in shadow.parameters():
                p.requires_grad_(False)
            self._qot_shadow = (shadow,)
        shadow = self._qot_shadow[0]
        shadow.train(self.training)
        shadow.allow_empty_lut = self.allow_empty_lut
        shadow.mlp[2].p = self.mlp[2].p
        params, buffers = padded.lightpath_params(self, cp)
        out = torch.func.functional_call(shadow, {**params, **buffers}, (data,))
        if self.training:
            padded.lightpath_copy_back(self, buffers, c, cp)
        return out

    def forward(self, data):
        if not self.conv1.bias.is_cuda:
            # a model left on the CPU: opt-in upload (QOT_AUTO_DEVICE=1), or a loud error -- never a CPU computation
            from . import auto_device
            if auto_device.enabled():
                return auto_device.forward(self, data)
            raise auto_device.cpu_model_error()
        if self._qot_cp is not None:
            return self._forward_padded(data)
        x, edge_index, batch = data.x, data.edge_index, data.batch
        n = x.shape[0]
        graph = graph_index_for(data, n, gat_self_loops=True, check=False)
        lut_idx = self._lut_rows(data, check=False)     # (the reference's LUT-less error comes after the layers: below)
        if graph.status_pending:
            # the one-launch index build's status word, read behind the host synchronisation the LUT rows cost anyway
            graph.status_pending = False
            check_index_status(x.device)
        lut_embedding = None
        pending = None             # (norm, raw conv output, BatchNorm partials) whose BatchNorm + ReLU the next projection applies
        for layer in range(1, self.num_layers + 1):
            conv, norm = getattr(self, f"conv{layer}"), getattr(self, f"norm{layer}")
            # (the projection's epilogue also leaves the layer's attention logits where a head is 128 channels wide)
            thin = pending is None and conv.thin_ok(x)      # raw features: the projection inside the attention kernels
            if thin:
                z, logits = None, None
            elif pending is None:
                z, logits = conv.project(x, with_logits=True)
            else:                  # relu(norm(x)) @ W^T with the normalised activations never written to memory
                pnorm, praw, ppart = pending
                if QF.logits_ok(conv.lin.out_features, conv.heads):
                    z, logits = pnorm.project_relu(praw, conv.lin.weight, partials=ppart, att=(conv.att_src, conv.att_dst))
                else:
                    z, logits = pnorm.project_relu(praw, conv.lin.weight, partials=ppart), None
                pending = None
            if self.training:      # the conv's epilogue leaves the BatchNorm's column partials behind
                raw, part = (conv.attend_thin(x, graph, bn_stats=True) if thin else
                             conv.attend(z, graph, bn_stats=True, logits=logits))
                partials = (part, conv.bias)
            else:
                raw, partials = (conv.attend_thin(x, graph) if thin else conv.attend(z, graph, logits=logits)), None
            width = raw.shape[1]
            # BatchNorm + ReLU folded into the NEXT projection's operand load (QF.BnLinearFn): the normalised activations are
            # never written to / read from HBM (one [N, 4C] tensor less resident per layer); the product runs on
            # csrc/gemm.hip's NT kernel.  Step time at cfg3 equals the library product + separate apply pass within noise
            # (34.9-35.2 vs 35.0-35.3 ms); QOT_FUSE_BN_PROJECTION=0 selects the latter.
            if (layer < self.num_layers and QF.gemm_ok(width, width) and raw.shape[0] > 0
                    and getattr(self, "_qot_fuse_bn_projection", os.environ.get("QOT_FUSE_BN_PROJECTION", "1") != "0")):
                pending = (norm, raw, partials)
            elif (layer == self.num_layers and lut_idx.numel() > 0 and raw.shape[0] > 0
                  and os.environ.get("QOT_NO_BN_ROWS", "0") != "1"):
                # the last BatchNorm + ReLU feeds nothing but the LUT rows: only those rows of it are formed (QF.BnRowsFn)
                lut_embedding = norm.rows(raw, to_i32(lut_idx), relu=True, partials=partials)
            else:
                x = norm(raw, relu=True, partials=partials)        # BatchNorm + F.relu fused, materialised
        idx = self._lut_rows(data)
        if idx.numel() == 0:          # only with allow_empty_lut: zero rows that still hang on the graph
            return x[:0, :self.mlp[3].out_features], batch[:0]
        if lut_embedding is None:
            lut_embedding = QF.RowsGatherFn.apply(x, to_i32(idx))
        lut_batch = batch.index_select(0, idx)
        l0, act, drop, l3 = self.mlp[0], self.mlp[1], self.mlp[2], self.mlp[3]
        h = QF.SmallLinearFn.apply(lut_embedding, l0.weight, l0.bias)       # head MLP (models.py:17-22,43)
        return QF.SmallLinearFn.apply(drop(act(h)), l3.weight, l3.bias), lut_batch
