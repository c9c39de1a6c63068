"""Device-side weight re-layouts: the packed operand images of the MLP kernels built ON THE DEVICE from the live parameters.

The host packers of the C ABI (`surf_sdf_pack_weights_{bf16,f16}`, `surf_sdf_smooth_pack_weights`,
`surf_blend_pack_weights_split`) are the definition of the layouts.  Each of them is a fixed gather: every packed value is
`c x[s]` for one source element `s` (a weight or bias of the effective matrices) and one constant `c` (1, 1/sqrt 2 for the
skip layer, log2 e for the blend network, 0 for padding), followed - for the split kernels - by the exact operand split.
A training step used to run them once per optimiser step: ~45 small device-to-host copies (each a stream sync), ~6 ms of
host packing, one upload.  Here the (s, c) map of a packer is RECOVERED ONCE by probing it with two inputs (x = 1, 2, 3 ...
and x = 1), after which a re-layout is `x[s] * c` + the split in a handful of torch ops on the device: no host round trip,
no sync.  `tests/test_host_modules.py::test_device_packing_*` checks byte identity with the host packers on random weights.
"""
import numpy as np
import torch

from . import _lib, ops

_SDF_SHAPES = [(128, 27), (128, 156), (101, 156), (128, 156), (128, 156), (128, 156), (129, 156)]
_MAPS = {}


# ---- reading values back out of a packed image ------------------------------------------------------------------------------

def _split_values(buf_u8, n_blocks, precision):
    """(n_blocks, NP, 64, 8) 16-bit pieces -> float64 (n_blocks, 64, 8): the pieces summed."""
    NP = {"bf16x3": 3, "f16x2": 2}[precision]
    u16 = buf_u8[:n_blocks * NP * 1024].view(np.uint16).reshape(n_blocks, NP, 64, 8)
    if precision == "bf16x3":
        vals = (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    else:
        vals = u16.view(np.float16).astype(np.float64)
    return vals.sum(axis=1)


class _Map:
    """idx (int64, -1 = structural zero) and coef (float64) per packed value, as device tensors on demand."""

    def __init__(self, idx, coef, **extra):
        self.idx_np, self.coef_np, self.extra = idx, coef, extra
        self._dev = {}

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            idx = torch.from_numpy(np.maximum(self.idx_np, 0)).to(device)
            self._dev[key] = (idx, torch.from_numpy(self.coef_np).to(device))
        return self._dev[key]


def _recover(y_ids, y_ones, n_src, exact_product):
    """Source index and coefficient of every packed value from the two probe runs; `exact_product(ids, c)` re-does the host
    arithmetic and must reproduce y_ids bit for bit (the packer really is this gather)."""
    coef = y_ones.astype(np.float64)
    idx = np.full(y_ids.shape, -1, dtype=np.int64)
    nz = coef != 0
    idx[nz] = np.rint(y_ids[nz] / coef[nz]).astype(np.int64) - 1
    assert (y_ids[~nz] == 0).all(), "a packed value with coefficient 0 is not zero"
    assert idx[nz].min() >= 0 and idx[nz].max() < n_src
    again = np.zeros_like(y_ids)
    again[nz] = exact_product((idx[nz] + 1).astype(np.float64), coef[nz])
    assert np.array_equal(again, y_ids), "the host packer is not the linear gather the probe assumes"
    return idx, coef


# ---- SDF network ------------------------------------------------------------------------------------------------------------

def _sdf_probe_layers(values):
    """[(W_l, b_l)] whose entries are `values` laid out flat in the order W_0, b_0, W_1, b_1, ..."""
    layers, o = [], 0
    for out_d, in_d in _SDF_SHAPES:
        W = values[o:o + out_d * in_d].reshape(out_d, in_d)
        o += out_d * in_d
        b = values[o:o + out_d]
        o += out_d
        layers.append((torch.from_numpy(np.ascontiguousarray(W)), torch.from_numpy(np.ascontiguousarray(b))))
    assert o == values.size
    return layers


def sdf_source_count():
    return sum(o * i + o for o, i in _SDF_SHAPES)


def _sdf_flat(isdf):
    """The effective (weight-normed) matrices and biases of SDFNetworkSparse as one flat device vector, W_0, b_0, W_1, ...
    (sdf_network.py:88-89: W = g v / |v|_row), computed on the device."""
    # one flat vector per parameter version: the split-bf16 image of the render kernels and the fp32 image of the training
    # kernels are both cut from it (44 small launches, twice per training step until round 5)
    ps = [p for l in range(7) for p in (getattr(isdf, f"lin{l}").weight_v, getattr(isdf, f"lin{l}").weight_g,
                                        getattr(isdf, f"lin{l}").bias)]
    key = tuple((p._version, p.data_ptr()) for p in ps)
    cached = getattr(isdf, "_surf_flat", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    parts = []
    for l in range(7):
        lin = getattr(isdf, f"lin{l}")
        v, g = lin.weight_v.detach().float(), lin.weight_g.detach().float()
        parts.append((v * (g / torch.linalg.norm(v, dim=1, keepdim=True))).reshape(-1))
        parts.append(lin.bias.detach().float().reshape(-1))
    flat = torch.cat(parts)
    object.__setattr__(isdf, "_surf_flat", (key, flat))
    return flat


def _sdf_split_map(precision):
    """(body map, tail map, NP, Scales<P>::W).  The logical order of the packed values - (k-step block, lane, k-slot) - is the
    same for both split policies, so the map is probed once with the bf16x3 packer (whose three pieces sum back exactly)."""
    # (the coefficients differ: see below)
    key = ("sdf",)
    if key not in _MAPS:
        L = _lib.lib()
        n = sdf_source_count()
        total = L.surf_sdf_bf16_packed_bytes()
        tail_floats = 164
        stream = total - tail_floats * 4
        n_blocks = stream // (3 * 1024)
        assert L.surf_sdf_f16_packed_bytes() == n_blocks * 2 * 1024 + tail_floats * 4
        runs = []
        for vals in (np.arange(1, n + 1, dtype=np.float32), np.ones(n, dtype=np.float32)):
            img = ops.sdf_pack_weights_split_host(_sdf_probe_layers(vals), "bf16x3")
            runs.append((_split_values(img, n_blocks, "bf16x3"), img[stream:].view(np.float32).astype(np.float64)))

        def prod(ids, c):                                            # host: float product h_W * scale (scale in {1, (float)(1/sqrt 2)})
            return (ids.astype(np.float32) * c.astype(np.float32)).astype(np.float64)
        _MAPS[key] = (_Map(*_recover(runs[0][0], runs[1][0], n, prod)), _Map(*_recover(runs[0][1], runs[1][1], n, prod)))
    body, tail = _MAPS[key]
    if precision == "f16x2":
        # The bf16x3 kernel runs in units of the softplus exponent (sdf_split_common.h, PolBf3::PRESCALED): its packer scales the
        # input columns and biases by c = 100 log2 e (c / sqrt 2 in the skip layer) and lin6's row 0 by ln 2 / 100; the f16x2
        # packer does not.  Same gather, those factors taken out again (exact float constants of the host packer).
        key2 = ("sdf", "f16x2")
        if key2 not in _MAPS:
            c = np.float64(np.float32(100.0 * 1.4426950408889634))
            cr = np.float64(np.float32(100.0 * 1.4426950408889634 / np.sqrt(2.0)))
            k = np.float64(np.float32(0.6931471805599453 / 100.0))
            r2 = np.float64(np.float32(1.0 / np.sqrt(2.0)))

            def plain(m):
                coef = m.coef_np.copy()
                coef[m.coef_np == c] = 1.0
                coef[m.coef_np == cr] = r2
                coef[m.coef_np == k] = 1.0
                assert set(np.unique(coef)) <= {0.0, 1.0, r2}, "unexpected scale in the bf16x3 weight image"
                return _Map(m.idx_np, coef)
            _MAPS[key2] = (plain(body), plain(tail))
        body, tail = _MAPS[key2]
    # Scales<P>::W: the exact power of two the f16x2 packer applies before the split
    return body, tail, {"bf16x3": 3, "f16x2": 2}[precision], {"bf16x3": 1.0, "f16x2": 256.0}[precision]


def _split_pieces(vals, precision):
    """The exact operand split of the kernels' host packers (split_host<P>), on the device: (..., ) float32 -> list of NP
    16-bit tensors."""
    if precision == "bf16x3":
        p0 = vals.to(torch.bfloat16)
        r = vals - p0.float()
        p1 = r.to(torch.bfloat16)
        p2 = (r - p1.float()).to(torch.bfloat16)
        return [p0, p1, p2]
    p0 = vals.to(torch.float16)
    return [p0, (vals - p0.float()).to(torch.float16)]


def sdf_pack_split_device(isdf, precision="bf16x3"):
    """== ops.sdf_pack_weights_split(state_dict, device, precision), byte for byte, without leaving the device."""
    body, tail, NP, w_scale = _sdf_split_map(precision)
    flat = _sdf_flat(isdf)
    dev = flat.device
    bi, bc = body.on(dev)
    vals = (flat[bi] * bc.float()) * w_scale                         # (n_blocks, 64, 8); padding: coefficient 0
    pieces = torch.stack(_split_pieces(vals, precision), dim=1).contiguous()      # (n_blocks, NP, 64, 8)
    ti, tc = tail.on(dev)
    tail_v = (flat[ti] * tc.float()).contiguous()
    return torch.cat([pieces.view(torch.uint8).reshape(-1), tail_v.view(torch.uint8).reshape(-1)])


def _sdf_smooth_map():
    key = ("smooth",)
    if key not in _MAPS:
        n = sdf_source_count()
        runs = [ops.sdf_smooth_pack_weights_host(_sdf_probe_layers(vals)).astype(np.float64)
                for vals in (np.arange(1, n + 1, dtype=np.float32), np.ones(n, dtype=np.float32))]

        def prod(ids, c):
            return (ids.astype(np.float32) * c.astype(np.float32)).astype(np.float64)
        _MAPS[key] = _Map(*_recover(runs[0], runs[1], n, prod))
    return _MAPS[key]


def sdf_pack_smooth_device(isdf):
    """== ops.sdf_smooth_pack_weights(state_dict, device) without leaving the device."""
    flat = _sdf_flat(isdf)
    idx, coef = _sdf_smooth_map().on(flat.device)
    return (flat[idx] * coef.float()).contiguous()


# ---- blending network ---------------------------------------------------------------------------------------------------------

_LOG2E = float(np.float32(1.44269504088896341))      # blend_split.hip: `constexpr float LOG2E`, widened to double in the products


def _blend_split_map():
    """(block map, tail map) of the split blend image; the logical order (block, lane, k-slot) is shared by the three layouts,
    probed once with the bf16x3 packer."""
    key = ("blend",)
    if key not in _MAPS:
        L = _lib.lib()
        n = L.surf_blend_raw_floats()
        n_blocks = 26
        a_bytes = n_blocks * 3 * 1024
        runs = []
        for vals in (np.arange(1, n + 1, dtype=np.float32), np.ones(n, dtype=np.float32)):
            img = ops.blend_pack_weights_split_host(vals, "bf16x3")
            runs.append((_split_values(img, n_blocks, "bf16x3"), img[a_bytes:].view(np.float32).astype(np.float64)))

        def prod(ids, c):
            # host: (float)((double) W * (double) c) with c = log2 e (weights, bias rows) or 1 (dot rows, scalars); the probe's
            # coefficient is that product at W = 1, i.e. (float) c
            cd = np.where(np.abs(c - np.float64(np.float32(_LOG2E))) < 1e-12, _LOG2E, c)
            return (ids * cd).astype(np.float32).astype(np.float64)
        _MAPS[key] = (_Map(*_recover(runs[0][0], runs[1][0], n, prod)), _Map(*_recover(runs[0][1], runs[1][1], n, prod)))
    return _MAPS[key]


def _blend_flat(color_network):
    cn = dict(color_network.named_parameters())
    flat = torch.cat([cn[k].detach().reshape(-1).float() for k in ops.BLEND_KEYS])
    flat = flat.clone()
    flat[0] = flat[0].abs()                                           # the image holds |s| (blending_network.py:80), raw index 0
    return flat


def _times(flat, idx, coef):
    """x[s] * c with the host packers' rounding: products by log2 e are formed in double and rounded once."""
    c32 = coef.float()
    is_log2e = (coef - float(np.float32(_LOG2E))).abs() < 1e-12
    x = flat[idx]
    return torch.where(is_log2e, (x.double() * _LOG2E).float(), x * c32)


def blend_pack_split_device(color_network, precision="bf16x3"):
    """== ops.blend_pack_weights(state_dict, device, precision=...) (a PackedBlend), without leaving the device."""
    body, tail = _blend_split_map()
    flat = _blend_flat(color_network)
    dev = flat.device
    bi, bc = body.on(dev)
    vals = _times(flat, bi, bc)
    if precision == "f32lds":
        blocks = vals.contiguous().view(torch.uint8).reshape(-1)
    else:
        blocks = torch.stack(_split_pieces(vals, precision), dim=1).contiguous().view(torch.uint8).reshape(-1)
    ti, tc = tail.on(dev)
    tail_v = _times(flat, ti, tc).contiguous().view(torch.uint8).reshape(-1)
    return ops.PackedBlend(torch.cat([blocks, tail_v]), precision)


def blend_raw_device(color_network):
    """The raw parameter buffer surf_blend_backward reads (state_dict order), on the device."""
    cn = dict(color_network.named_parameters())
    return torch.cat([cn[k].detach().reshape(-1).float() for k in ops.BLEND_KEYS]).contiguous()


def supported(sdf_precision, blend_precision):
    return sdf_precision in ("bf16x3", "f16x2") and blend_precision in ("bf16x3", "f16x2", "f32lds")

