"""Host-side mirror of models/modules/reg_network.py (SparseCostRegNet / SparseCostRegNetList).

Parameter names follow torchsparse 2.1's modules inside the reference's blocks so that the reference's
checkpoints load: ``nets.{s}.conv{i}.net.0.kernel`` (27, C_in, C_out), ``nets.{s}.conv{i}.net.1.{weight,bias,
running_mean,running_var,num_batches_tracked}``, ``nets.{s}.out_lin.weight``.  The arithmetic runs in
``surf_spconv`` (csrc/spconv.hip).  torchsparse itself is absent here (third party): the convolution
semantics are those documented in ``oracle.surf_oracle.sparse_unet`` -- PARITY UNPINNED.
Eval-mode BatchNorm (running statistics) is folded into the convolution epilogues; train mode uses batch statistics
(`surf_bn_train_affine`, forward only) and updates the running ones like torch.
"""
import math

import torch
import torch.nn as nn

from . import ops


class _SpConv3d(nn.Module):
    def __init__(self, inc, outc, transposed=False):
        super().__init__()
        self.kernel = nn.Parameter(torch.zeros(27, inc, outc))
        std = 1.0 / math.sqrt((outc if transposed else inc) * 27)   # torchsparse Conv3d.reset_parameters
        with torch.no_grad():
            self.kernel.uniform_(-std, std)


class _Block(nn.Module):
    """BasicSparseConvolutionBlock / BasicSparseDeconvolutionBlock: net = [Conv3d, BatchNorm, ReLU]."""

    def __init__(self, inc, outc, stride=1, transposed=False):
        super().__init__()
        self.stride, self.transposed = stride, transposed
        self.net = nn.Sequential(_SpConv3d(inc, outc, transposed), nn.BatchNorm1d(outc), nn.ReLU(True))

    def bn_affine(self):
        bn = self.net[1]
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias.detach() - bn.running_mean * scale
        return scale.float().contiguous(), shift.float().contiguous()

    def prepared(self, use_mfma=True):
        """(kernel, bn scale, bn shift, split-bf16 operand image or None), cached until a parameter or buffer changes
        (optimiser step, load_state_dict, .to(device)): saves five small launches per convolution and the repacking.
        The running statistics are also written by surf_bn_train_affine through raw pointers, which torch's version
        counters do not see: the train path sets `_bn_dirty`."""
        conv, bn = self.net[0], self.net[1]
        wkey = (conv.kernel._version, conv.kernel.data_ptr(), bool(use_mfma))
        if getattr(self, "_wprep", None) is None or self._wprep[0] != wkey:
            w = conv.kernel.detach().float().contiguous()
            self._wprep = (wkey, w, ops.spconv_pack_weights(w) if use_mfma else None)
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        bkey = tuple((t._version, t.data_ptr()) for t in ts)
        if getattr(self, "_bprep", None) is None or self._bprep[0] != bkey or getattr(self, "_bn_dirty", False):
            self._bprep = (bkey,) + self.bn_affine()
            self._bn_dirty = False
        return self._wprep[1], self._bprep[1], self._bprep[2], self._wprep[2]


class SparseCostRegNet(nn.Module):
    def __init__(self, d_in, d_out=8, d_base=8, down_rule="dilate"):
        super().__init__()
        if down_rule not in ops.DOWN_RULES:
            raise ValueError(f"reg_network.down_rule must be one of {sorted(ops.DOWN_RULES)}, got {down_rule!r}")
        self.down_rule = down_rule
        self.use_mfma = True      # wide layers on the matrix cores (spconv_mfma.hip); False: every layer on spconv.hip
        if d_base != 8 or d_out != 8 or d_in not in (8, 16):
            raise NotImplementedError("surf_spconv is instantiated for d_base = d_out = 8, d_in in {8, 16} (confs/*.conf)")
        b = d_base
        self.conv0 = _Block(d_in, b)
        self.conv1 = _Block(b, 2 * b, stride=2)
        self.conv2 = _Block(2 * b, 2 * b)
        self.conv3 = _Block(2 * b, 4 * b, stride=2)
        self.conv4 = _Block(4 * b, 4 * b)
        self.conv5 = _Block(4 * b, 8 * b, stride=2)
        self.conv6 = _Block(8 * b, 8 * b)
        self.conv7 = _Block(8 * b, 4 * b, stride=2, transposed=True)
        self.conv9 = _Block(4 * b, 2 * b, stride=2, transposed=True)
        self.conv11 = _Block(2 * b, b, stride=2, transposed=True)
        self.out_lin = nn.Linear(b, d_out, bias=False)

    def _conv(self, blk, x, table, out_coords, mode, skip=None):
        w, scale, shift, packed = blk.prepared(self.use_mfma)
        if self.training:   # batch statistics (forward only: nothing here is differentiable), running statistics updated
            raw = ops.spconv(x, table, out_coords, mode, w, None, None, None, packed=packed)
            blk._bn_dirty = True
            return ops.bn_train_relu(raw, blk.net[1], skip)
        return ops.spconv(x, table, out_coords, mode, w, scale, shift, skip, packed=packed)

    def forward(self, feats, coords, D, table=None):
        """feats (N, d_in) fp32, coords (N,3) int32 on the D lattice -> (out (N,8), mid (N,8))  (reg_network.py:69-88)"""
        t0 = table if table is not None else ops.table_from_coords(coords, D)
        c0 = self._conv(self.conv0, feats, t0, coords, ops.SUBM)
        cd1, t1, D1 = ops.down_sites(coords, D, self.down_rule)
        x = self._conv(self.conv1, c0, t0, cd1, ops.DOWN)
        c2 = self._conv(self.conv2, x, t1, cd1, ops.SUBM)
        cd2, t2, D2 = ops.down_sites(cd1, D1, self.down_rule)
        x = self._conv(self.conv3, c2, t1, cd2, ops.DOWN)
        c4 = self._conv(self.conv4, x, t2, cd2, ops.SUBM)
        cd3, t3, D3 = ops.down_sites(cd2, D2, self.down_rule)
        x = self._conv(self.conv5, c4, t2, cd3, ops.DOWN)
        x = self._conv(self.conv6, x, t3, cd3, ops.SUBM)
        x = self._conv(self.conv7, x, t3, cd2, ops.UP, skip=c4)
        x = self._conv(self.conv9, x, t2, cd1, ops.UP, skip=c2)
        x = self._conv(self.conv11, x, t1, coords, ops.UP, skip=c0)
        out = ops.row_linear8(x, self.out_lin.weight.detach().float().contiguous())
        return out, x


class SparseCostRegNetList(nn.Module):
    def __init__(self, confs):
        super().__init__()
        d_in, d_out, d_base = confs.get_list("d_in"), confs.get_list("d_out"), confs.get_list("d_base")
        self.num_stages = len(d_in)
        # optional key (ours): which stride-2 output-site rule of torchsparse the checkpoint was trained with
        rule = confs.get_string("down_rule", "dilate")
        self.nets = nn.ModuleList([SparseCostRegNet(d_in[i], d_out[i], d_base[i], rule) for i in range(self.num_stages)])

    def forward(self, feats, coords, D, stage_idx, table=None):
        return self.nets[stage_idx](feats, coords, D, table)
