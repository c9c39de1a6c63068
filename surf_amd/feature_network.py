"""Host-side mirror of models/modules/feature_network.py FeatureNetwork (same parameter names:
``encoder_layers.i.{0,1}.conv.weight``, ``decoder_layers.i.conv.weight``, ``out_layers.i.weight``).
The convolutions / InstanceNorm run in csrc/fpn.hip on NHWC activations; the four 4-channel outputs are
produced directly as texel4 maps (nv,H,W,4), returned coarse -> fine like the reference (:178)."""
import torch
import torch.nn as nn

from . import ops
from .grads import accumulate


class _Conv(nn.Module):
    """Conv2d wrapper of feature_network.py:6-25 (conv without bias + InstanceNorm + ReLU): parameters only."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.stride = stride
        self.conv = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


class _Deconv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False)


def _layout(w, tag, build):
    """A re-laid-out copy of the convolution weight `w`, cached on the parameter per version: built once per optimiser step (the
    frozen matching FPN: once per refresh) instead of once per forward AND backward call - ~60 small launches a training step,
    the backward's share of them on its device-bound main chain."""
    cache = w.__dict__.setdefault("_surf_layouts", {})
    key = (w._version, w.data_ptr())
    hit = cache.get(tag)
    if hit is None or hit[0] != key or not ops.layout_cache:
        with torch.no_grad():
            hit = cache[tag] = (key, build(w.detach().float()))
    return hit[1]


def _pack_conv(w):
    """(Cout, Cin, 3, 3) -> [ky][kx][Cin (padded to a multiple of 4)][Cout]."""
    def build(v):
        v = v.permute(2, 3, 1, 0)
        cin = v.shape[2]
        if cin % 4:
            v = torch.cat([v, torch.zeros(3, 3, 4 - cin % 4, v.shape[3], device=v.device)], dim=2)
        return v.contiguous()
    return _layout(w, "conv", build)


def _pack_deconv(w):
    """(Cin, Cout, 3, 3) -> [ky][kx][Cin][Cout]."""
    return _layout(w, "deconv", lambda v: v.permute(2, 3, 0, 1).contiguous())


class FeatureNetwork(nn.Module):
    def __init__(self, confs):
        super().__init__()
        d_in = confs.get_int("d_in")
        d_base = confs.get_int("d_base")
        d_outs = confs.get_list("d_out")
        if d_in != 3 or d_base != 8 or list(d_outs) != [4, 4, 4, 4]:
            raise NotImplementedError("surf_conv3x3 is instantiated for d_in 3, d_base 8, d_out [4,4,4,4] (confs/*.conf)")
        self.num_stage = len(d_outs)
        self.encoder_layers = nn.ModuleList()
        self.decoder_layers = nn.ModuleList()
        self.out_layers = nn.ModuleList()
        c = d_in
        for i in range(self.num_stage):
            m = d_base * 2 ** i
            self.encoder_layers.append(nn.Sequential(_Conv(c, m, 2 if i > 0 else 1), _Conv(m, m, 1)))
            c = m
            self.out_layers.append(nn.Conv2d(m, d_outs[i], 3, 1, 1, bias=False))
            if i < self.num_stage - 1:
                self.decoder_layers.append(_Deconv(d_base * 2 ** (i + 1), m))

    def forward(self, imgs, tape=None):
        """imgs (nv,3,H,W) in [0,1) -> list of 4 texel4 maps (nv,h,w,4), coarse -> fine.  tape: a list that receives what
        `backward` needs (layer inputs, raw convolution outputs, InstanceNorm statistics)."""
        if imgs.shape[-2] % 8 or imgs.shape[-1] % 8:
            raise ValueError("image height and width must be divisible by 8")
        x = ops.pack_texel4(imgs.detach().float().contiguous())
        rec = tape is not None
        # the layers with C_in >= 16 run on the matrix cores (csrc/fpn_mfma.hip): fp32-equivalent, or - in train mode under the
        # bf16 policy (conf key train_precision) - on bf16-rounded operands; inference is always fp32-equivalent
        prec = ops.colgram_precision if self.training else 0

        def norm(y, skip=None):
            if not rec:
                return ops.inorm_relu_(y, skip=skip), None, None
            out, stats = ops.inorm_relu_(y, skip=skip, want_stats=True, in_place=False)    # y stays: the raw convolution output
            return out, y, stats

        enc, enc_rec = [], []
        for i in range(self.num_stage):
            for blk in self.encoder_layers[i]:
                x_in = x
                x = ops.conv3x3(x, _pack_conv(blk.conv.weight), blk.conv.out_channels, blk.stride, prec)
                x, raw, stats = norm(x)
                enc_rec.append(dict(blk=blk, x_in=x_in, raw=raw, stats=stats))
            enc.append(x)
        d = enc[-1]
        dec = [None] * self.num_stage
        dec[-1] = d
        dec_rec = [None] * self.num_stage
        for i in range(self.num_stage - 2, -1, -1):
            blk = self.decoder_layers[i]
            d_in = d
            d = ops.deconv3x3_s2(d, _pack_deconv(blk.conv.weight), blk.conv.out_channels, prec)
            d, raw, stats = norm(d, skip=enc[i])
            dec_rec[i] = dict(blk=blk, x_in=d_in, raw=raw, stats=stats)
            dec[i] = d
        outs = [ops.conv3x3(dec[i], _pack_conv(self.out_layers[i].weight), 4, 1, prec) for i in range(self.num_stage)]
        if rec:
            tape.append(dict(enc=enc_rec, dec=dec_rec, dec_out=dec, prec=prec))
        return outs[::-1]

    def backward(self, tape, g_outs_c2f, sink=None):
        """Reverse sweep: g_outs_c2f = gradients of the four texel4 outputs (coarse -> fine, like forward's return value).
        ACCUMULATES into the `.grad` of every convolution weight (or into `sink`, a grads.GradSink).  Input gradients reuse the forward kernels (stride 1: the
        flipped, transposed kernel; stride 2 <-> transposed convolution), weight gradients `surf_conv3x3_wgrad`, the
        InstanceNorm + ReLU (+ skip) backward `surf_bn_relu_backward` per view."""
        def acc(p, g):
            accumulate(p, g, sink)

        def flipT(w):
            """Conv2d weight (Cout, Cin, 3, 3) -> the packed kernel [ky][kx][Cout][Cin] of its input gradient (stride 1)."""
            return _layout(w, "flipT", lambda v: v.flip(2, 3).permute(2, 3, 0, 1).contiguous())

        t = tape[-1]
        n = self.num_stage
        prec = t.get("prec", 0)                                                 # the policy the forward ran under
        on_side = ops.side.active("fpn") and g_outs_c2f[0].is_cuda
        deferred = []

        def wgrad(a, b, stride, finish, w):
            """The weight gradient of one layer: a leaf of this sweep - on the side stream beside the input-gradient chain
            (ops.SideStream); `finish` maps the kernel's [ky][kx][.][.] layout to the parameter's, applied after the join."""
            if on_side:
                dw = ops.side.run(lambda: ops.conv3x3_wgrad(a, b, stride, prec), keep=(a, b))
                deferred.append((w, dw, finish))
            else:
                acc(w, finish(ops.conv3x3_wgrad(a, b, stride, prec)))
        g_outs = g_outs_c2f[::-1]                                               # index i = level i (0 = finest)
        d_dec = []
        for i in range(n):
            w = self.out_layers[i].weight
            g = g_outs[i].contiguous()
            d_dec.append(ops.conv3x3(g, flipT(w), w.shape[1], 1, prec))
            wgrad(t["dec_out"][i], g, 1, lambda dw: dw.permute(3, 2, 0, 1), w)  # [ky][kx][ci][co]
        d_enc = [None] * n
        for i in range(n - 1):                                                  # dec[i] = IN(deconv(dec[i+1])) + enc[i]
            r = t["dec"][i]
            w = r["blk"].conv.weight                                            # (Cin, Cout, 3, 3)
            g = d_dec[i]
            d_enc[i] = g
            d_raw = ops.inorm_relu_backward(r["raw"], g, r["stats"])
            # input gradient of the transposed convolution = stride-2 convolution with [ky][kx][co][ci]
            d_dec[i + 1] = d_dec[i + 1] + ops.conv3x3(d_raw, _layout(w, "deconv_dgrad", lambda v: v.permute(2, 3, 1, 0).contiguous()), w.shape[0], 2, prec)
            wgrad(d_raw, r["x_in"], 2, lambda dw: dw.permute(3, 2, 0, 1), w)    # [ky][kx][co][ci]
        d_enc[n - 1] = d_dec[n - 1]
        g = d_enc[n - 1]
        for k in range(2 * n - 1, -1, -1):                                      # encoder blocks, last to first
            i, j = k // 2, k % 2
            r = t["enc"][k]
            w = r["blk"].conv.weight                                            # (Cout, Cin, 3, 3)
            d_raw = ops.inorm_relu_backward(r["raw"], g.contiguous(), r["stats"])
            wgrad(r["x_in"], d_raw, r["blk"].stride, lambda dw, ci=w.shape[1]: dw[:, :, :ci].permute(3, 2, 0, 1), w)  # [ky][kx][ci (padded)][co]
            if k == 0:
                break
            if r["blk"].stride == 1:
                g = ops.conv3x3(d_raw, flipT(w), w.shape[1], 1, prec)
            else:                                                               # stride-2 conv <- transposed conv, [ky][kx][co][ci]
                g = ops.deconv3x3_s2(d_raw, _layout(w, "conv_s2_dgrad", lambda v: v.permute(2, 3, 0, 1).contiguous()), w.shape[1], prec)
            if j == 0:                                                          # entering stage i - 1's output: add its skip gradient
                g = g + d_enc[i - 1]
        if on_side:                                                             # the weight gradients: meet the side stream here
            ops.side.join(lanes=(0,))
            for w, dw, finish in deferred:
                acc(w, finish(dw))
