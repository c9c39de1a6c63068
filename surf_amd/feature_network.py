"""Host-side mirror of models/modules/feature_network.py FeatureNetwork (same parameter names:
``encoder_layers.i.{0,1}.conv.weight``, ``decoder_layers.i.conv.weight``, ``out_layers.i.weight``).
The convolutions / InstanceNorm run in csrc/fpn.hip on NHWC activations; the four 4-channel outputs are
produced directly as texel4 maps (nv,H,W,4), returned coarse -> fine like the reference (:178)."""
import torch
import torch.nn as nn

from . import ops


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


def _pack_conv(w):
    """(Cout, Cin, 3, 3) -> [ky][kx][Cin (padded to a multiple of 4)][Cout]."""
    w = w.detach().float().permute(2, 3, 1, 0)
    cin = w.shape[2]
    if cin % 4:
        w = torch.cat([w, torch.zeros(3, 3, 4 - cin % 4, w.shape[3], device=w.device)], dim=2)
    return w.contiguous()


def _pack_deconv(w):
    """(Cin, Cout, 3, 3) -> [ky][kx][Cin][Cout]."""
    return w.detach().float().permute(2, 3, 0, 1).contiguous()


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

    def forward(self, imgs):
        """imgs (nv,3,H,W) in [0,1) -> list of 4 texel4 maps (nv,h,w,4), coarse -> fine."""
        if imgs.shape[-2] % 8 or imgs.shape[-1] % 8:
            raise ValueError("image height and width must be divisible by 8")
        x = ops.pack_texel4(imgs.detach().float().contiguous())
        enc = []
        for i in range(self.num_stage):
            for blk in self.encoder_layers[i]:
                x = ops.conv3x3(x, _pack_conv(blk.conv.weight), blk.conv.out_channels, blk.stride)
                ops.inorm_relu_(x)
            enc.append(x)
        d = enc[-1]
        dec = [None] * self.num_stage
        dec[-1] = d
        for i in range(self.num_stage - 2, -1, -1):
            blk = self.decoder_layers[i]
            d = ops.deconv3x3_s2(d, _pack_deconv(blk.conv.weight), blk.conv.out_channels)
            ops.inorm_relu_(d, skip=enc[i])
            dec[i] = d
        outs = [ops.conv3x3(dec[i], _pack_conv(self.out_layers[i].weight), 4, 1) for i in range(self.num_stage)]
        return outs[::-1]
