"""Host-side mirror of models/modules/reg_network.py (SparseCostRegNet / SparseCostRegNetList).

Parameter names follow torchsparse 2.1's modules inside the reference's blocks so that the reference's
checkpoints load: ``nets.{s}.conv{i}.net.0.kernel`` (27, C_in, C_out), ``nets.{s}.conv{i}.net.1.{weight,bias,
running_mean,running_var,num_batches_tracked}``, ``nets.{s}.out_lin.weight``.  The arithmetic runs in
``surf_spconv`` (csrc/spconv.hip).  torchsparse itself is absent here (third party): the convolution
semantics are those documented in ``oracle.surf_oracle.sparse_unet`` -- PARITY UNPINNED.
Eval-mode BatchNorm (running statistics) is folded into the convolution epilogues; train mode uses batch statistics
(`surf_bn_train_affine`) and updates the running ones like torch; a train-mode forward can record a tape for
`SparseCostRegNet.backward` (HIP kernels throughout: no autograd graph).
"""
import math

import torch
import torch.nn as nn

from . import ops
from .grads import accumulate


KERNEL_ORDERS = ("xfast", "zfast")
TRANSPOSED_PAIRINGS = ("same", "mirrored")


def slice_permutation(kernel_order="xfast", mirrored=False):
    """perm (27,) with  kernel_as_the_HIP_kernels_walk_it[k] = checkpoint_kernel[perm[k]].  The kernels enumerate the 27 offsets
    with x fastest and z slowest (csrc/spconv.hip) and a transposed layer's slice k serves offset k; which enumeration torchsparse
    2.1.0 stores, and which slice its transposed layers pair with which offset, cannot be established here (third party, absent:
    PARITY UNPINNED), so both are conf keys like `down_rule` - a host-side permutation of the slices, no kernel change:
    reg_network.kernel_order        xfast (default) | zfast (checkpoint slice (x+1) 9 + (y+1) 3 + (z+1))
    reg_network.transposed_pairing  same (default)  | mirrored (an up layer's slice k serves offset -k, i.e. slice 26 - k)
    Both permutations are involutions and commute, so the composed one is its own inverse (used for the weight gradient)."""
    if kernel_order not in KERNEL_ORDERS:
        raise ValueError(f"reg_network.kernel_order must be one of {KERNEL_ORDERS}, got {kernel_order!r}")
    perm = []
    for z in (-1, 0, 1):
        for y in (-1, 0, 1):
            for x in (-1, 0, 1):
                a, b, c = (-x, -y, -z) if mirrored else (x, y, z)
                perm.append((a + 1) * 9 + (b + 1) * 3 + (c + 1) if kernel_order == "zfast" else (c + 1) * 9 + (b + 1) * 3 + (a + 1))
    return perm


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
        self._perm = None           # slice_permutation of this block (None = identity), SparseCostRegNet.set_conventions
        self._perm_dev = None

    def set_slice_permutation(self, perm):
        self._perm = None if perm is None or list(perm) == list(range(27)) else list(perm)
        self._perm_dev = None
        self._wprep = self._tprep = None

    def slice_index(self, device):
        """The permutation as a device index tensor (None = identity)."""
        if self._perm is None:
            return None
        if self._perm_dev is None or self._perm_dev.device != device:
            self._perm_dev = torch.tensor(self._perm, dtype=torch.long, device=device)
        return self._perm_dev

    def walk_kernel(self):
        """The kernel in the order the HIP kernels walk it: checkpoint slices permuted by the torchsparse conventions in force."""
        w = self.net[0].kernel.detach().float()
        idx = self.slice_index(w.device)
        return (w if idx is None else w.index_select(0, idx)).contiguous()

    def bn_affine(self):
        bn = self.net[1]
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias.detach() - bn.running_mean * scale
        return scale.float().contiguous(), shift.float().contiguous()

    def prepared(self, use_mfma=True, affine=True, thin=False):
        """(kernel, bn scale, bn shift, split-bf16 operand image or None), cached until a parameter or buffer changes
        (optimiser step, load_state_dict, .to(device)): saves five small launches per convolution and the repacking.
        The running statistics are also written by surf_bn_train_affine through raw pointers, which torch's version
        counters do not see: the train path sets `_bn_dirty`.  affine=False (train mode: batch statistics, the folded
        running-statistics affine is not used): scale / shift are None and nothing is computed for them."""
        conv, bn = self.net[0], self.net[1]
        wkey = (conv.kernel._version, conv.kernel.data_ptr(), bool(use_mfma), bool(thin))
        if getattr(self, "_wprep", None) is None or self._wprep[0] != wkey:
            w = self.walk_kernel()
            self._wprep = (wkey, w, ops.spconv_pack_weights(w, thin) if use_mfma else None)
        if not affine:
            return self._wprep[1], None, None, self._wprep[2]
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        bkey = tuple((t._version, t.data_ptr()) for t in ts)
        if getattr(self, "_bprep", None) is None or self._bprep[0] != bkey or getattr(self, "_bn_dirty", False):
            self._bprep = (bkey,) + self.bn_affine()
            self._bn_dirty = False
        return self._wprep[1], self._bprep[1], self._bprep[2], self._wprep[2]

    def prepared_dgrad(self, mode, use_mfma=True, thin=False):
        """ops.dgrad_weights of this block's kernel (transposed / mirrored slices + their split-bf16 image), cached per
        parameter version: one re-layout per optimiser step instead of one per backward call."""
        conv = self.net[0]
        key = (conv.kernel._version, conv.kernel.data_ptr(), int(mode), bool(use_mfma), bool(thin))
        if getattr(self, "_tprep", None) is None or self._tprep[0] != key:
            self._tprep = (key, ops.dgrad_weights(self.walk_kernel(), mode, use_mfma, thin))
        return self._tprep[1]


class SparseCostRegNet(nn.Module):
    def __init__(self, d_in, d_out=8, d_base=8, down_rule=ops.DEFAULT_DOWN_RULE, kernel_order="xfast", transposed_pairing="same"):
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
        self.set_conventions(kernel_order, transposed_pairing)

    def blocks(self):
        return [getattr(self, f"conv{i}") for i in (0, 1, 2, 3, 4, 5, 6, 7, 9, 11)]

    def set_conventions(self, kernel_order="xfast", transposed_pairing="same"):
        """Which slice enumeration / transposed pairing of torchsparse the stored kernels follow (slice_permutation): takes effect
        at the next forward (the weight re-layout caches are dropped); the parameters themselves stay in checkpoint order."""
        if transposed_pairing not in TRANSPOSED_PAIRINGS:
            raise ValueError(f"reg_network.transposed_pairing must be one of {TRANSPOSED_PAIRINGS}, got {transposed_pairing!r}")
        self.kernel_order, self.transposed_pairing = kernel_order, transposed_pairing
        for blk in self.blocks():
            blk.set_slice_permutation(slice_permutation(kernel_order, blk.transposed and transposed_pairing == "mirrored"))

    def _conv(self, blk, x, in_site, out_site, mode, skip=None, tape=None, counters=None):
        """One block on x (rows of `in_site` = (table, coords)) -> rows of `out_site`.  tape: a list that receives what
        `backward` needs (train mode only: eval mode folds the BN into the convolution epilogue and keeps no raw output)."""
        # train_precision = bf16: every layer on the matrix cores with bf16-rounded operands, one product per offset - the thin
        # layers of the finest lattices too (round 6: their weights are only packed under this policy)
        thin = ops.thin_mfma == "all" or (ops.thin_mfma == "bf16" and self.training and ops.colgram_precision == 1)
        w, scale, shift, packed = blk.prepared(self.use_mfma, affine=not self.training, thin=thin)
        if self.training:   # batch statistics, running statistics updated
            raw = ops.spconv(x, in_site[0], out_site[1], mode, w, None, None, None, packed=packed,
                             bf16=ops.colgram_precision == 1)
            blk._bn_dirty = True
            saved = {} if tape is not None else None
            # bf16 policy: 16-channel rows get a bf16 shadow in the same pass; the (16 -> 8) convolutions gather from it
            shadow = ops.colgram_precision == 1 and ops.bf16_rows and raw.shape[1] == 16
            y = ops.bn_train_relu(raw, blk.net[1], skip, saved, counters=counters, shadow=shadow)
            if tape is not None:
                tape.append(dict(blk=blk, x=x, raw=raw, y=y, skip=skip, in_site=in_site, out_site=out_site, mode=mode, w=w, **saved))
                # the backward's re-laid-out kernel (cached per parameter version) is built HERE: the forward has idle bubbles
                # behind its voxel-count reads, the backward sweep is device bound and would pay these launches on its main chain
                if ops.layout_cache:
                    blk.prepared_dgrad(mode, self.use_mfma, thin=ops.thin_mfma == "all" or
                                       (ops.thin_mfma == "bf16" and ops.colgram_precision == 1))
            return y
        if tape is not None:
            raise RuntimeError("SparseCostRegNet: the backward tape is recorded in train mode only")
        return ops.spconv(x, in_site[0], out_site[1], mode, w, scale, shift, skip, packed=packed)

    def forward(self, feats, coords, D, table=None, tape=None):
        """feats (N, d_in) fp32, coords (N,3) int32 on the D lattice -> (out (N,8), mid (N,8))  (reg_network.py:69-88)"""
        q1 = q2 = q3 = None
        if self.down_rule == "pad0":
            # uncentred stride-2 window 2q + {0,1,2}^3 (ops.down_sites): every level's coordinates are stored + 1, which turns
            # it into the centred window the kernels walk; the true lattices shrink as (D - 3) // 2 + 1
            coords = (coords + 1).contiguous()
            # (lattices below 15^3 run out of sites before the third level: those levels are EMPTY, as in the oracle's
            # down_coords - nothing flows through their blocks, the transposed convolutions above them add zeros)
            q1 = max((D - 3) // 2 + 1, 0)
            q2 = max((q1 - 3) // 2 + 1, 0)
            q3 = max((q2 - 3) // 2 + 1, 0)
            D, table = D + 1, None
        counters = [] if self.training else None       # the blocks' num_batches_tracked, bumped together below
        t0 = table if table is not None else ops.table_from_coords(coords, D)
        s0 = (t0, coords)
        c0 = self._conv(self.conv0, feats, s0, s0, ops.SUBM, tape=tape, counters=counters)
        cd1, t1, D1 = ops.down_sites(coords, D, self.down_rule, q1)
        s1 = (t1, cd1)
        x = self._conv(self.conv1, c0, s0, s1, ops.DOWN, tape=tape, counters=counters)
        c2 = self._conv(self.conv2, x, s1, s1, ops.SUBM, tape=tape, counters=counters)
        cd2, t2, D2 = ops.down_sites(cd1, D1, self.down_rule, q2)
        s2 = (t2, cd2)
        x = self._conv(self.conv3, c2, s1, s2, ops.DOWN, tape=tape, counters=counters)
        c4 = self._conv(self.conv4, x, s2, s2, ops.SUBM, tape=tape, counters=counters)
        cd3, t3, D3 = ops.down_sites(cd2, D2, self.down_rule, q3)
        s3 = (t3, cd3)
        x = self._conv(self.conv5, c4, s2, s3, ops.DOWN, tape=tape, counters=counters)
        x = self._conv(self.conv6, x, s3, s3, ops.SUBM, tape=tape, counters=counters)
        x = self._conv(self.conv7, x, s3, s2, ops.UP, skip=c4, tape=tape, counters=counters)
        x = self._conv(self.conv9, x, s2, s1, ops.UP, skip=c2, tape=tape, counters=counters)
        x = self._conv(self.conv11, x, s1, s0, ops.UP, skip=c0, tape=tape, counters=counters)
        if counters:
            torch._foreach_add_(counters, 1)
        out = ops.row_linear8(x, self.out_lin.weight.detach().float().contiguous())
        if tape is not None:
            tape.append(dict(lin=True, x=x, feats=feats))
        return out, x

    def backward(self, tape, d_out, d_mid=None, sink=None, pending=None):
        """Reverse sweep over a tape recorded by a train-mode forward: gradients of sum(out d_out) + sum(mid d_mid) are
        ACCUMULATED into the `.grad` (or into `sink`, a grads.GradSink) of every convolution kernel, BatchNorm weight / bias and
        out_lin.weight; returns the
        gradient of `feats` (N, d_in).  Each block: BatchNorm(batch statistics) + ReLU + skip backward (surf_bn_relu_backward),
        then the convolution's input gradient as a sparse convolution on the swapped lattices and its kernel gradient
        (ops.spconv_backward).  The kernel gradients are leaves of the sweep: launched on the side stream (ops.SideStream) and
        accumulated after the join - here, or, with `pending` (a list), by the caller: it receives a closure to call after ITS
        ops.side.join()."""
        def acc(p, g):
            accumulate(p, g, sink)

        last = tape[-1]
        assert last.get("lin"), "tape: recorded by SparseCostRegNet.forward(..., tape=[])"
        x11, feats = last["x"], last["feats"]
        W = self.out_lin.weight.detach().float().contiguous()
        grads = {}

        def add(t, g):
            k = id(t)
            grads[k] = g if k not in grads else grads[k] + g

        on_side = ops.side.active("unet") and d_out.is_cuda
        kernel_grads = []
        d_out = d_out.float().contiguous()
        add(x11, ops.row_linear8(d_out, W.t().contiguous()))            # d_out @ W
        if d_mid is not None:
            add(x11, d_mid.float().contiguous())
        if on_side:                                                      # d_out^T x11: a leaf too (side stream, accumulated after the join)
            g_lin = ops.side.run(lambda: ops.colgram(d_out, x11), lane=0, keep=(d_out, x11))
        else:
            g_lin = None
            acc(self.out_lin.weight, ops.colgram(d_out, x11))
        for e in reversed(tape[:-1]):
            g = grads.pop(id(e["y"]), None)
            if g is None:
                continue
            if e["skip"] is not None:
                add(e["skip"], g)
            if e["raw"].shape[0] == 0:        # an empty level (tiny lattices): nothing flows through this block
                continue
            draw, dgamma, dbeta = ops.bn_relu_backward(e["raw"], g.contiguous(), e["scale"], e["shift"], e["stats"], train=True,
                                                       shadow=ops.colgram_precision == 1 and ops.bf16_rows and e["raw"].shape[1] == 16)
            bn = e["blk"].net[1]
            acc(bn.weight, dgamma)
            acc(bn.bias, dbeta)
            dx, dW = ops.spconv_backward(e["x"], e["in_site"][0], e["in_site"][1], e["out_site"][0], e["out_site"][1], e["mode"],
                                         e["w"], draw, use_mfma=self.use_mfma,
                                         dgrad=e["blk"].prepared_dgrad(e["mode"], self.use_mfma,
                                                                       thin=ops.thin_mfma == "all" or
                                                                       (ops.thin_mfma == "bf16" and ops.colgram_precision == 1)),
                                         on_side=on_side)
            kernel_grads.append((e["blk"], dW))
            add(e["x"], dx)
        # the kernel gradients were launched on the side stream (leaves of this sweep: they overlap its input-gradient chain)
        def finish():
            if g_lin is not None:
                acc(self.out_lin.weight, g_lin)
            for blk, dW in kernel_grads:
                idx = blk.slice_index(dW.device)           # back to checkpoint slice order (the permutation is an involution)
                acc(blk.net[0].kernel, dW if idx is None else dW.index_select(0, idx))

        if on_side and pending is not None:                # the caller joins the side stream later (more to overlap with)
            pending.append(finish)
        else:
            if on_side:
                ops.side.join(lanes=(0,))
            finish()
        return grads.pop(id(feats))


def _has(confs, key):
    try:
        return key in confs
    except TypeError:
        return confs.get_string(key, None) is not None


class SparseCostRegNetList(nn.Module):
    def __init__(self, confs):
        super().__init__()
        d_in, d_out, d_base = confs.get_list("d_in"), confs.get_list("d_out"), confs.get_list("d_base")
        self.num_stages = len(d_in)
        # optional key (ours): which stride-2 output-site rule of torchsparse the checkpoint was trained with
        rule = confs.get_string("down_rule", ops.DEFAULT_DOWN_RULE)
        # optional keys (ours): the two other torchsparse conventions a loaded checkpoint depends on (slice_permutation)
        order = confs.get_string("kernel_order", "xfast")
        pairing = confs.get_string("transposed_pairing", "same")
        self.conventions_named = all(_has(confs, k) for k in ("down_rule", "kernel_order", "transposed_pairing"))
        self.nets = nn.ModuleList([SparseCostRegNet(d_in[i], d_out[i], d_base[i], rule, order, pairing)
                                   for i in range(self.num_stages)])
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._warn_unnamed_conventions())

    def conventions(self):
        """The torchsparse conventions in force, as the conf keys a checkpoint's conf should carry (they are NOT in the
        state_dict: its keys are the reference's, so that checkpoints stay loadable on both sides)."""
        n = self.nets[0]
        return {"down_rule": n.down_rule, "kernel_order": n.kernel_order, "transposed_pairing": n.transposed_pairing}

    def set_conventions(self, down_rule=None, kernel_order=None, transposed_pairing=None):
        cur = self.conventions()
        if down_rule is not None and down_rule not in ops.DOWN_RULES:
            raise ValueError(f"reg_network.down_rule must be one of {sorted(ops.DOWN_RULES)}, got {down_rule!r}")
        for n in self.nets:
            n.down_rule = down_rule or cur["down_rule"]
            n.set_conventions(kernel_order or cur["kernel_order"], transposed_pairing or cur["transposed_pairing"])
        self.conventions_named = True

    def _warn_unnamed_conventions(self):
        if not self.conventions_named and not getattr(self, "_warned", False):
            import warnings
            self._warned = True
            warnings.warn("SparseCostRegNetList: a checkpoint was loaded but the conf names none / not all of reg_network.down_rule, "
                          f".kernel_order, .transposed_pairing; using {self.conventions()} (PARITY UNPINNED: torchsparse 2.1.0's "
                          "conventions are recollections; checkpoints trained with surf_amd before round 5 used down_rule = dilate). "
                          "Name the three keys in the conf that travels with the checkpoint; scripts/dtu_chamfer.py --sweep "
                          "finds the combination that reproduces a reference Chamfer.", stacklevel=3)

    def forward(self, feats, coords, D, stage_idx, table=None, tape=None):
        return self.nets[stage_idx](feats, coords, D, table, tape)
