"""Torch-facing wrappers of the C-ABI kernels (device memory and streams are PyTorch's; the math is HIP).

Every function checks dtype / device / contiguity, allocates outputs with torch and launches on the
current stream.  Nothing here computes on the CPU except the one-off weight re-layouts (host code of
the library) and 4x4 camera inversions.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

BLEND_KEYS = ["s"] + [f"{m}.{i}.{p}" for m, idxs in (("ray_dir_fc", (0, 2)), ("base_fc", (0, 2)), ("vis_fc", (0, 2)),
                                                      ("vis_fc2", (0, 2)), ("rgb_fc", (0, 2, 4)))
                      for i in idxs for p in ("weight", "bias")]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# bench.py: a list that receives (name, start event, end event, algorithmic units) of the instrumented backward launches
# (HIP events on the launch stream); None = no timing
kernel_events = None


class _timed:
    """with _timed("costvol_bwd", units): ... - brackets the launches inside with a HIP event pair when a bench asked for it."""

    def __init__(self, name, units=0):
        self.name, self.units = name, units

    def __enter__(self):
        if kernel_events is not None:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if kernel_events is not None and exc[0] is None:
            self.b.record()
            kernel_events.append((self.name, self.a, self.b, self.units))
        return False


class SideStream:
    """The backward sweep of a training step on several HIP streams.  Its kernels work on the step's 65 k samples or wait on row
    gathers / an LDS hash: latency bound, a fraction of the chip's wave slots each - so branches that do not depend on one
    another run side by side (users, each an A/B switch; profiles/r06_side_streams.txt):
      unet    the sparse U-Nets' kernel gradients (leaves of the sweep) beside their input-gradient chain          lane 0
      render  the colour branch (lane 0) and the smooth (H.1) branch (lane 1) beside the SDF value / gradient branch
      match   the matching chain (matching-field backward -> densify backward, fine -> coarse) ahead of the U-Net / cost-volume
              chain it feeds, one event per stage                                                                 lane 2
      costvol the stages' cost-volume backward (a leaf: only the parent-feature scatter feeds the next stage)           lane 2 (reused)
      loss    the 2 n photometric terms of the loss, both directions (autograd._PhotometricMulti; lanes 4..7; measured: a loss of
              1.5 ms - the launches are bound by the memory system, not by latency; off by default)
      fpn     the FPN's weight gradients (measured: a loss - those launches fill the chip; off by default)
      fwd     the forward's smooth / random-point branches and the frozen matching FPN (measured: no gain; off by default)
    `with side.fork(lane): launch(...)` orders the lane after everything issued so far on the current stream; `run(fn)` does that
    and marks the results for the current stream; `keep()` holds operands until `join()` (the caching allocator reuses a block
    for the CURRENT stream as soon as its last reference dies); `join()` makes the current stream wait for every open lane -
    before the results are read.  Never inside a lane: a synchronising host read (it starves the other streams).
    SURF_SIDE_STREAM=0: in-order launches."""

    def __init__(self):
        # "0" none, "1" the users that measured a gain (profiles/r06_side_streams.txt), "all", or a comma list of users
        env = os.environ.get("SURF_SIDE_STREAM", "1")
        self.enabled = env != "0"
        self.users = {"0": set(), "1": {"unet", "render", "match", "costvol"}, "all": None}.get(env, set(env.split(",")))
        self.priority = int(os.environ.get("SURF_SIDE_PRIORITY", "0"))      # of the lanes' streams (0 = as the current stream's)
        self._streams = {}
        self._keep = {}          # (device, lane) -> tensors held until that lane is joined
        self._open = set()

    def active(self, user=None):
        # never while a bench is bracketing launches with event pairs (per-kernel times want in-order launches)
        return (self.enabled and kernel_events is None and torch.cuda.is_available()
                and (self.users is None or user is None or user in self.users))

    def fork(self, lane=0):
        cur = torch.cuda.current_stream()
        key = (cur.device.index, lane)
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = torch.cuda.Stream(device=cur.device, priority=self.priority)
        st.wait_stream(cur)
        self._open.add(key)
        return torch.cuda.stream(st)

    def lane_stream(self, lane, device=None):
        """The stream object of `lane` on `device` (default: the current one), created on first use, NOT ordered after anything:
        for graph nodes that LIVE on a lane - autograd runs a node's backward on the stream its forward ran on and orders the
        gradients that cross streams itself."""
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        key = (idx, lane)
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = torch.cuda.Stream(device=torch.device("cuda", idx), priority=self.priority)
        return st

    def on_lane(self, lane):
        """True when the current stream is the stream of `lane`."""
        st = self._streams.get((torch.cuda.current_device(), lane))
        return st is not None and torch.cuda.current_stream() == st

    def run(self, fn, lane=0, keep=()):
        """fn() on side stream `lane`; its result tensors (a tensor, or a list / tuple / dict of them, nested) are allocated in
        that stream's pool and will be read - and freed - on the current one: marked with record_stream."""
        main = torch.cuda.current_stream()
        with self.fork(lane):
            out = fn()

        def claim(o):
            if torch.is_tensor(o):
                if o.is_cuda:
                    o.record_stream(main)
            elif isinstance(o, dict):
                for v in o.values():
                    claim(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    claim(v)

        claim(out)
        self.keep(*keep, lane=lane)
        return out

    @staticmethod
    def mark():
        """An event on the CURRENT stream (inside `with fork(lane)`: that lane) for `wait_for` on another stream."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    @staticmethod
    def wait_for(event):
        torch.cuda.current_stream().wait_event(event)

    def keep(self, *tensors, lane=0):
        """Hold `tensors` (operands a lane reads, allocated on another stream) until that lane is joined."""
        if torch.cuda.is_available():
            key = (torch.cuda.current_device(), lane)
            self._keep.setdefault(key, []).extend(t for t in tensors if t is not None)

    def join(self, lanes=None):
        """The current stream waits for the open lanes of its device (all of them, or `lanes`)."""
        if not self._open:
            self._keep.clear()
            return
        cur = torch.cuda.current_stream()
        for key in sorted(self._open):
            if key[0] == cur.device.index and (lanes is None or key[1] in lanes):
                cur.wait_stream(self._streams[key])
                self._open.discard(key)
                self._keep.pop(key, None)


side = SideStream()
# The runtime multiplexes HIP streams onto a handful of hardware queues (4 by default): which streams share one decides how much really
# overlaps, and a fifth stream (or RCCL's own) reshuffles the assignment.  The cost-volume backward therefore REUSES the matching
# chain's lane (which ran ahead and is idle by then) instead of opening another stream: main + three lanes in all.  Measured on the
# default line's step (ms, group present / absent, two runs each): own lane 67.3 65.1 / 65.3 64.3, smooth lane 67.2 64.5 / 66.1 66.0,
# matching lane 65.5 63.5 / 64.5 64.5; under DistributedDataParallel 69.9 70.9 | 70.6 66.2 | 65.2 66.3.
# graph nodes that live on the matching lane (the depth tap, the loss's photometric node): autograd runs their backward there (A/B: 0 =
# the tap on the main stream forks the lane itself, the photometric terms run on the main stream)
lane_nodes = os.environ.get("SURF_LANE_NODES", "1") != "0"
COSTVOL_LANE = int(os.environ.get("SURF_COSTVOL_LANE", "2"))
SMOOTH_LANE = int(os.environ.get("SURF_SMOOTH_LANE", "1"))       # the render backward's smooth branch (A/B: 0 = behind the colour branch)


class _ZeroPool:
    """Small zero-initialised buffers carved out of one pre-zeroed block per device: a training step asks for ~100 of them
    (the 27 x C_in x C_out weight-gradient accumulators of the sparse U-Net, 1.2 MB a stage) and each torch.zeros is a ~5 us
    fill launch on a stream that is never idle.  A slice is handed out ONCE (never recycled: it lives as long as its views do),
    the block is replaced when it runs out."""
    BLOCK = 4 << 20          # floats (16 MB)
    LIMIT = 1 << 18          # larger requests fill for themselves

    def __init__(self):
        self._blocks = {}

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        if n > self.LIMIT or n == 0:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        key = (device.type, device.index)
        blk = self._blocks.get(key)
        step = (n + 63) & ~63   # 256-byte slices
        if blk is None or blk[1] + step > self.BLOCK:
            blk = [torch.zeros(self.BLOCK, dtype=torch.float32, device=device), 0]
            self._blocks[key] = blk
        out = blk[0][blk[1]:blk[1] + n].view(shape)
        blk[1] += step
        return out


_zero_pool = _ZeroPool()


def small_zeros(shape, device):
    """fp32 zeros(shape) on `device`, from the pool above when small."""
    return _zero_pool.take(tuple(shape), torch.device(device))


def _p(t):
    t = getattr(t, "tensor", t)            # PackedBlend
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _chk(t, dtype, name, dev=True):
    if not torch.is_tensor(t) or t.dtype != dtype or not t.is_contiguous():
        raise TypeError(f"{name}: expected a contiguous {dtype} tensor")
    if dev and not t.is_cuda:
        raise TypeError(f"{name}: expected a GPU tensor (the SuRF hot path has no CPU fallback)")
    return t


def _host_f32(t):
    return np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ------------------------------------------------------------------------------------------------
# layout helpers
# ------------------------------------------------------------------------------------------------


def pack_texel4(x):
    """(n, C<=4, H, W) fp32 NCHW -> (n, H, W, 4) texel4 (surf_pack_texel4)."""
    _chk(x, torch.float32, "x")
    n, C, H, W = x.shape
    out = torch.empty(n, H, W, 4, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().surf_pack_texel4(_p(x), n, C, H, W, _p(out), _stream()), "surf_pack_texel4")
    return out


class SparseVolumes:
    """Per-stage sparse feature rows + dense int32 index tables, ordered fine -> coarse (surf.py:159)."""

    def __init__(self, feats, tables):
        self.vols, self.tables, self.dims = [], [], []
        for f, t in zip(feats, tables):
            if f.shape[1] != 8 or f.shape[0] == 0:   # the kernels read row 0 (with weight 0) for empty corners
                f8 = torch.zeros(max(f.shape[0], 1), 8, dtype=torch.float32, device=f.device)
                f8[:, :f.shape[1]] = f
                f = f8
            if t.dtype != torch.int32:
                t = t.to(torch.int32)
            self.vols.append(_chk(f.contiguous(), torch.float32, "sparse volume"))
            self.tables.append(_chk(t.contiguous(), torch.int32, "index table"))
            assert t.dim() == 3 and t.shape[0] == t.shape[1] == t.shape[2], "index tables must be cubic"
            self.dims.append(int(t.shape[0]))
        self.n = len(self.vols)
        self._vp = _ptr_array(self.vols)
        self._tp = _ptr_array(self.tables)
        self._dp = (ctypes.c_int * self.n)(*self.dims)


def sdf_effective_weights(sd, prefix="implicit_surface.sdf_network."):
    """weight_norm(dim=0) re-parameterisation (sdf_network.py:88-89): W = g * v / ||v||_row."""
    out = []
    for l in range(7):
        if f"{prefix}lin{l}.weight_v" in sd:
            v = sd[f"{prefix}lin{l}.weight_v"].detach().to("cpu", torch.float32)
            g = sd[f"{prefix}lin{l}.weight_g"].detach().to("cpu", torch.float32)
            W = v * (g / torch.linalg.norm(v, dim=1, keepdim=True))
        else:
            W = sd[f"{prefix}lin{l}.weight"].detach().to("cpu", torch.float32)
        out.append((W.contiguous(), sd[f"{prefix}lin{l}.bias"].detach().to("cpu", torch.float32).contiguous()))
    return out


def sdf_pack_weights_host(layers):
    """[(W_l, b_l)] effective matrices -> packed numpy buffer (surf_sdf_pack_weights, host code)."""
    shapes = [(128, 27), (128, 156), (101, 156), (128, 156), (128, 156), (128, 156)]
    for l, (W, b) in enumerate(layers):
        if l < 6 and tuple(W.shape) != shapes[l]:
            raise NotImplementedError(
                f"lin{l}.weight has shape {tuple(W.shape)}; the HIP SDF kernel implements the shipped architecture "
                "(d_hidden 128, n_layers 6, skip_in [3], multires 4, feat_channels 28)")
    if layers[6][0].shape[1] != 156:
        raise NotImplementedError("lin6 must take 156 inputs")
    Ws = [_host_f32(W) for W, _ in layers]
    bs = [_host_f32(b) for _, b in layers]
    L = _lib.lib()
    out = np.zeros(L.surf_sdf_packed_floats(), dtype=np.float32)
    wp = (ctypes.c_void_p * 7)(*[w.ctypes.data for w in Ws])
    bp = (ctypes.c_void_p * 7)(*[b.ctypes.data for b in bs])
    _lib.check(L.surf_sdf_pack_weights(wp, bp, _np_ptr(out)), "surf_sdf_pack_weights")
    return out


SDF_PRECISIONS = ("f32", "bf16x3", "f16x2")
# the blend kernels of the library (render.blend_precision): "f32" = blend.hip (fp32 MFMA, register-resident weights
# stream, round 1); the others = blend_split.hip (weights resident in LDS, two wavefronts per SIMD): "f32lds" fp32 MFMA,
# "bf16x3" exact three-way bf16 split, "f16x2" two fp16 pieces
BLEND_PRECISIONS = ("f32", "f32lds", "bf16x3", "f16x2")
BLEND_DEFAULT = "bf16x3"
_BLEND_ID = {"bf16x3": 1, "f16x2": 2, "f32lds": 3}   # SURF_BLEND_BF16X3 / SURF_BLEND_F16X2 / SURF_BLEND_F32
_SPLIT_ABI = {"bf16x3": "bf16", "f16x2": "f16"}   # precision -> infix of the C-ABI entry points


def sdf_pack_weights_split_host(layers, precision):
    """[(W_l, b_l)] effective matrices -> byte stream of split 16-bit weights (surf_sdf_pack_weights_{bf16,f16})."""
    sdf_pack_weights_host(layers)  # shape validation only
    infix = _SPLIT_ABI[precision]
    Ws = [_host_f32(W) for W, _ in layers]
    bs = [_host_f32(b) for _, b in layers]
    L = _lib.lib()
    out = np.zeros(getattr(L, f"surf_sdf_{infix}_packed_bytes")(), dtype=np.uint8)
    wp = (ctypes.c_void_p * 7)(*[w.ctypes.data for w in Ws])
    bp = (ctypes.c_void_p * 7)(*[b.ctypes.data for b in bs])
    _lib.check(getattr(L, f"surf_sdf_pack_weights_{infix}")(wp, bp, _np_ptr(out)), f"surf_sdf_pack_weights_{infix}")
    return out


def sdf_pack_weights_bf16_host(layers):
    return sdf_pack_weights_split_host(layers, "bf16x3")


def sdf_pack_weights_split(sd, device, prefix="implicit_surface.sdf_network.", precision="bf16x3"):
    return torch.from_numpy(sdf_pack_weights_split_host(sdf_effective_weights(sd, prefix), precision)).to(device)


def sdf_pack_weights_bf16(sd, device, prefix="implicit_surface.sdf_network."):
    return sdf_pack_weights_split(sd, device, prefix, "bf16x3")


def sdf_packed_precision(packed):
    """Which SDF kernel a packed weight tensor belongs to (fp32 tensors: the fp32 kernel; byte streams: by size)."""
    if packed.dtype == torch.float32:
        return "f32"
    if packed.dtype == torch.uint8:
        for precision, infix in _SPLIT_ABI.items():
            if packed.numel() == getattr(_lib.lib(), f"surf_sdf_{infix}_packed_bytes")():
                return precision
    raise ValueError("packed SDF weights: not an output of surf_sdf_pack_weights / _bf16 / _f16")


def sdf_pack_weights(sd, device, prefix="implicit_surface.sdf_network."):
    return torch.from_numpy(sdf_pack_weights_host(sdf_effective_weights(sd, prefix))).to(device)


def blend_raw_weights(sd, prefix="implicit_surface.color_network."):
    return np.concatenate([_host_f32(sd[prefix + k]).reshape(-1) for k in BLEND_KEYS])


def blend_pack_weights_host(raw):
    L = _lib.lib()
    if raw.size != L.surf_blend_raw_floats():
        raise NotImplementedError("colour network shape differs from the shipped BlendingNetwork(d_feature=16)")
    out = np.zeros(L.surf_blend_packed_floats(), dtype=np.float32)
    _lib.check(L.surf_blend_pack_weights(_np_ptr(np.ascontiguousarray(raw, dtype=np.float32)), _np_ptr(out)),
               "surf_blend_pack_weights")
    return out


def blend_pack_weights_split_host(raw, precision):
    """LDS image of the split blend kernels (surf_blend_pack_weights_split): 16-bit weight pieces + bias / dot rows."""
    L = _lib.lib()
    if raw.size != L.surf_blend_raw_floats():
        raise NotImplementedError("colour network shape differs from the shipped BlendingNetwork(d_feature=16)")
    pid = _BLEND_ID[precision]
    out = np.zeros(L.surf_blend_split_packed_bytes(pid), dtype=np.uint8)
    _lib.check(L.surf_blend_pack_weights_split(_np_ptr(np.ascontiguousarray(raw, dtype=np.float32)), _np_ptr(out), pid),
               "surf_blend_pack_weights_split")
    return out


class PackedBlend:
    """The LDS image of a split blend kernel together with the precision it was laid out for.  The f16x2 and f32lds images have
    the SAME size, so the layout cannot be recovered from the bytes: it travels with them explicitly (an attribute on the
    tensor, as in round 2, is lost by clone() / to() / a state-dict round trip and the f32lds image then ran through the f16x2
    kernel without any error)."""
    __slots__ = ("tensor", "precision")

    def __init__(self, tensor, precision):
        if precision not in _BLEND_ID:
            raise ValueError(f"blend precision must be one of {sorted(_BLEND_ID)}")
        if tensor.dtype != torch.uint8 or tensor.numel() != _lib.lib().surf_blend_split_packed_bytes(_BLEND_ID[precision]):
            raise ValueError(f"not a {precision} image of surf_blend_pack_weights_split")
        self.tensor, self.precision = tensor, precision

    def to(self, device):
        return PackedBlend(self.tensor.to(device), self.precision)

    @property
    def device(self):
        return self.tensor.device


def blend_pack_weights(sd, device, prefix="implicit_surface.color_network.", precision="f32"):
    """Packed BlendingNetwork weights for the blend kernel of `precision` (fp32 tensor: blend.hip; PackedBlend: blend_split.hip)."""
    raw = blend_raw_weights(sd, prefix)
    if precision == "f32":
        return torch.from_numpy(blend_pack_weights_host(raw)).to(device)
    return PackedBlend(torch.from_numpy(blend_pack_weights_split_host(raw, precision)).to(device), precision)


def blend_packed_precision(packed):
    """Which blend kernel a packed weight object belongs to.  Raw byte tensors are accepted only where the size identifies
    the layout; an ambiguous size raises instead of guessing."""
    if isinstance(packed, PackedBlend):
        return packed.precision
    if packed.dtype == torch.float32:
        return "f32"
    if packed.dtype == torch.uint8:
        hits = [precision for precision, pid in _BLEND_ID.items() if packed.numel() == _lib.lib().surf_blend_split_packed_bytes(pid)]
        if len(hits) == 1:
            return hits[0]
        if len(hits) > 1:
            raise ValueError(f"a {packed.numel()}-byte blend image fits the layouts {hits}: pass the PackedBlend that "
                             "blend_pack_weights returned (it carries the precision)")
    raise ValueError("packed blend weights: not an output of surf_blend_pack_weights / surf_blend_pack_weights_split")


# ------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------


_linspace_cache = {}


_nf_cache = {}


def _near_fars_host(near_fars):
    """The (nv, 2) near / far planes as a host array (the kernels take them by value).  A device tensor costs a synchronising
    copy: paid once per tensor (keyed by storage address + version), not once per stage and direction - and never from inside
    a side-stream section of the backward sweep, where a host stall starves the other streams (ops.SideStream)."""
    if not near_fars.is_cuda:
        return np.ascontiguousarray(near_fars.detach().to(torch.float32).numpy())
    key = (near_fars.data_ptr(), near_fars._version, tuple(near_fars.shape), near_fars.dtype)
    hit = _nf_cache.get(key)
    if hit is None:
        if len(_nf_cache) >= 16:
            _nf_cache.clear()
        # the tensor itself is kept: a live tensor's address cannot be handed to another one
        hit = _nf_cache[key] = (near_fars, np.ascontiguousarray(near_fars.detach().to("cpu", torch.float32).numpy()))
    return hit[1]


def _linspace_dev(lo, hi, n, dev):
    """torch.linspace computed on the CPU - like the reference, whose sample fractions are CPU linspaces moved to the device
    (implicit_surface.py:271, matching_field.py:31) - and kept on the device: one host-to-device copy per (lo, hi, n), not one
    per call (a pageable copy is a synchronising operation)."""
    key = (float(lo), float(hi), int(n), str(dev))
    t = _linspace_cache.get(key)
    if t is None:
        t = torch.linspace(float(lo), float(hi), int(n), dtype=torch.float32).to(dev)
        _linspace_cache[key] = t
    return t


def ray_setup(rays_o, rays_d, near, far, mvol, volumes, n_samples, sample_ranges, n_depth, want_z=False, jitter=None):
    """implicit_surface.py:268-311 + :72-86.  Returns dict(mid_z, dists, pts, vmask[, z_vals]).
    jitter: None (render.perturb = 0) or (R, n_stage) float32 on the device = the `torch.rand([R, 1]) - 0.5` draws of
    render.perturb > 0, one column per stage."""
    R = rays_o.shape[0]
    S = int(sum(n_samples))
    dev = rays_o.device
    _chk(rays_o, torch.float32, "rays_o")
    _chk(rays_d, torch.float32, "rays_d")
    near = _chk(near.reshape(-1).contiguous(), torch.float32, "near")
    far = _chk(far.reshape(-1).contiguous(), torch.float32, "far")
    _chk(mvol, torch.float32, "matching volume")
    assert near.numel() == R and far.numel() == R
    if jitter is not None:
        _chk(jitter, torch.float32, "jitter")
        assert tuple(jitter.shape) == (R, len(n_samples)), jitter.shape
    lin_depth = _linspace_dev(0.0, 1.0, n_depth, dev)
    key = ("stages", tuple(int(n) for n in n_samples), str(dev))
    lin_s = _linspace_cache.get(key)
    if lin_s is None:
        lin_s = _linspace_cache[key] = torch.cat([torch.linspace(0.0, 1.0, int(n), dtype=torch.float32) for n in n_samples]).to(dev)
    out = {
        "mid_z": torch.empty(R, S, dtype=torch.float32, device=dev),
        "dists": torch.empty(R, S, dtype=torch.float32, device=dev),
        "pts": torch.empty(R * S, 3, dtype=torch.float32, device=dev),
        "vmask": torch.empty(R * S, dtype=torch.uint8, device=dev),
    }
    if want_z:
        out["z_vals"] = torch.empty(R, S, dtype=torch.float32, device=dev)
    ns = (ctypes.c_int * len(n_samples))(*[int(n) for n in n_samples])
    rg = (ctypes.c_float * len(sample_ranges))(*[float(r) for r in sample_ranges])
    rc = _lib.lib().surf_ray_setup(_p(rays_o), _p(rays_d), _p(near), _p(far), R, _p(mvol), int(mvol.shape[-1]),
                                   _p(lin_depth), int(n_depth), _p(lin_s), ns, rg, len(n_samples), _p(jitter),
                                   ctypes.c_float(2.0 / n_samples[0]), volumes._tp, volumes._dp, volumes.n,
                                   _p(out.get("z_vals")), _p(out["mid_z"]), _p(out["dists"]), _p(out["pts"]),
                                   _p(out["vmask"]), _stream())
    _lib.check(rc, "surf_ray_setup")
    return out


_scratch_cache = {}


def _sdf_scratch(n, device, precision="f32"):
    fn = "surf_sdf_scratch_bytes" if precision == "f32" else f"surf_sdf_{_SPLIT_ABI[precision]}_scratch_bytes"
    need = getattr(_lib.lib(), fn)(int(n))
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _scratch_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _scratch_cache[key] = buf
    return buf


def sdf_mlp(pts, volumes, packed, mask=None, want_grad=True, compact_active=True, active_idx=None, active_count=None):
    """sdf_network.py:95-141 at n points.  Returns (sdf (n,), grad (n,3) or None); masked-out rows are
    left at sdf=100 / grad=0 (what render_core substitutes, implicit_surface.py:93,99).
    active_count (with active_idx from compact_counted, split kernels only): the number of valid entries of active_idx as a
    DEVICE tensor - the kernel reads it there, no host sync."""
    _chk(pts, torch.float32, "pts")
    precision = sdf_packed_precision(packed)    # which kernel the packed weights were laid out for
    _chk(packed, torch.float32 if precision == "f32" else torch.uint8, "packed weights")
    n = pts.shape[0]
    dev = pts.device
    if mask is not None:
        _chk(mask, torch.uint8, "mask")
        sdf = torch.full((n,), 100.0, dtype=torch.float32, device=dev)
        grad = torch.zeros(n, 3, dtype=torch.float32, device=dev) if want_grad else None
    else:
        sdf = torch.empty(n, dtype=torch.float32, device=dev)
        grad = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_grad else None
    idx, n_eval = None, n
    if active_idx is not None:
        idx, n_eval = active_idx, int(active_idx.shape[0])
        if n_eval == 0:
            return sdf, grad
    elif mask is not None and compact_active:
        idx = compact(mask)                 # wavefront tiles of active points only (one host sync for the count)
        n_eval = int(idx.shape[0])
        if n_eval == 0:
            return sdf, grad
    scratch = _sdf_scratch(n_eval, dev, precision) if want_grad else None
    name = {"f32": "surf_sdf_mlp", "bf16x3": "surf_sdf_mlp_bf16x3", "f16x2": "surf_sdf_mlp_f16x2"}[precision]
    if active_count is not None:
        if precision == "f32" or active_idx is None:
            raise ValueError("active_count needs active_idx and one of the split kernels (bf16x3 / f16x2)")
        _chk(active_count, torch.int32, "active_count")
        fn = getattr(_lib.lib(), name + "_dn")
        rc = fn(_p(pts), _p(idx), n_eval, _p(active_count), volumes._vp, volumes._tp, volumes._dp, volumes.n, _p(packed), _p(sdf),
                _p(grad), _p(scratch), _stream())
        _lib.check(rc, name + "_dn")
        return sdf, grad
    fn = getattr(_lib.lib(), name)
    rc = fn(_p(pts), _p(None if idx is not None else mask), _p(idx), n_eval, volumes._vp, volumes._tp, volumes._dp, volumes.n,
            _p(packed), _p(sdf), _p(grad), _p(scratch), _stream())
    _lib.check(rc, name)
    return sdf, grad


def sdf_lattice(axes, volumes, packed, out, x0, nx, sign=-1.0):
    """out[x0:x0+nx] (a slab of the (X, Y, Z) lattice tensor `out`) = sign * sdf on the lattice axes[0][x0:x0+nx] x axes[1] x
    axes[2], by the forward-only split kernel in lattice mode (no point tensor: implicit_surface.py:337-351's meshgrid / cat
    is the kernel's index arithmetic).  Split precisions only."""
    precision = sdf_packed_precision(packed)
    if precision == "f32":
        raise ValueError("sdf_lattice: bf16x3 / f16x2 weights only (the fp32-MFMA kernel takes point tensors)")
    for ax in axes:
        _chk(ax, torch.float32, "lattice axis")
    _chk(out, torch.float32, "lattice values")
    ny, nz = int(axes[1].shape[0]), int(axes[2].shape[0])
    assert tuple(out.shape) == (int(axes[0].shape[0]), ny, nz) and 0 <= x0 and x0 + nx <= out.shape[0]
    name = f"surf_sdf_lattice_{precision}"
    rc = getattr(_lib.lib(), name)(_p(axes[0][x0:]), _p(axes[1]), _p(axes[2]), int(nx), ny, nz, volumes._vp, volumes._tp, volumes._dp,
                                   volumes.n, _p(packed), _p(out[x0:]), ctypes.c_float(sign), _stream())
    _lib.check(rc, name)


def sdf_smooth_pack_weights_host(layers):
    """[(W_l, b_l)] effective matrices -> fp32 image of the SDF network for surf_sdf_smooth (both orientations of every matrix)."""
    sdf_pack_weights_host(layers)  # shape validation only
    Ws = [_host_f32(W) for W, _ in layers]
    bs = [_host_f32(b) for _, b in layers]
    L = _lib.lib()
    out = np.zeros(L.surf_sdf_smooth_packed_floats(), dtype=np.float32)
    wp = (ctypes.c_void_p * 7)(*[w.ctypes.data for w in Ws])
    bp = (ctypes.c_void_p * 7)(*[b.ctypes.data for b in bs])
    _lib.check(L.surf_sdf_smooth_pack_weights(wp, bp, _np_ptr(out)), "surf_sdf_smooth_pack_weights")
    return out


def sdf_smooth_pack_weights(sd, device, prefix="implicit_surface.sdf_network."):
    return torch.from_numpy(sdf_smooth_pack_weights_host(sdf_effective_weights(sd, prefix))).to(device)


def sdf_smooth(pts, volumes, packed, active_idx=None, want_grad=False):
    """sdf_network.py:143-152: smooth = d/dx sum_a (d sdf/d x_a) at n points.  Returns (smooth (n,3), grad (n,3) or
    None); rows not in active_idx stay zero (implicit_surface.py:100-103)."""
    _chk(pts, torch.float32, "pts")
    _chk(packed, torch.float32, "packed weights")
    n = pts.shape[0]
    if active_idx is not None:
        smooth = torch.zeros(n, 3, dtype=torch.float32, device=pts.device)
        grad = torch.zeros(n, 3, dtype=torch.float32, device=pts.device) if want_grad else None
        n_eval = int(active_idx.shape[0])
        if n_eval == 0:
            return smooth, grad
    else:
        smooth = torch.empty(n, 3, dtype=torch.float32, device=pts.device)
        grad = torch.empty(n, 3, dtype=torch.float32, device=pts.device) if want_grad else None
        n_eval = n
    rc = _lib.lib().surf_sdf_smooth(_p(pts), _p(active_idx), n_eval, volumes._vp, volumes._tp, volumes._dp, volumes.n,
                                    _p(packed), _p(grad), _p(smooth), _stream())
    _lib.check(rc, "surf_sdf_smooth")
    return smooth, grad


def sdf_backward(pts, ybar, gbar, volumes, packed, want_dvols=True, dvols=None):
    """Gradients of sum_n (ybar_n sdf_n + gbar_n . grad_n) w.r.t. the EFFECTIVE (weight-normed) matrices / biases of
    lin0..lin6 and the sparse feature rows (surf_sdf_backward; the batch reductions dW = adj^T in are surf_colgram_p: matrix cores).
    packed: sdf_smooth_pack_weights.  Returns {"weight": [7 tensors shaped like W_l], "bias": [7], "volumes": [per level (N_s,8)]}."""
    _chk(pts, torch.float32, "pts")
    _chk(ybar, torch.float32, "ybar")
    _chk(gbar, torch.float32, "gbar")
    _chk(packed, torch.float32, "packed weights")
    n, dev = pts.shape[0], pts.device
    in_v = torch.empty(7, n, 160, dtype=torch.float32, device=dev)
    in_d = torch.empty(7, n, 160, dtype=torch.float32, device=dev)
    tb = torch.empty(6, n, 128, dtype=torch.float32, device=dev)
    tdb = torch.empty(6, n, 128, dtype=torch.float32, device=dev)
    if dvols is not None:       # rows to ACCUMULATE into (the kernel adds with atomics either way)
        assert len(dvols) == volumes.n and all(d.shape == v.shape and d.is_contiguous() for d, v in zip(dvols, volumes.vols))
    else:
        dvols = [torch.zeros_like(v) for v in volumes.vols] if want_dvols else None
    with _timed("sdf_bwd", n):
        rc = _lib.lib().surf_sdf_backward(_p(pts), _p(ybar), _p(gbar), n, volumes._vp, volumes._tp, volumes._dp, volumes.n,
                                          _ptr_array(dvols) if dvols is not None else None, _p(packed), _p(in_v), _p(in_d), _p(tb),
                                          _p(tdb), _stream())
    _lib.check(rc, "surf_sdf_backward")
    shapes = [(128, 27), (128, 156), (101, 156), (128, 156), (128, 156), (128, 156), (129, 156)]
    dW, db = [], []
    for l in range(6):
        nl, kl = shapes[l]
        full = colgram(tb[l][:, :nl], in_v[l][:, :kl], with_sum=True)            # (N_l, K_l + 1): [tb^T in | sum tb]
        db.append(full[:, kl].contiguous())
        full = colgram(tdb[l][:, :nl], in_d[l][:, :kl], with_sum=True, out=full)  # + tdb^T in'   (its sum column is unused)
        dW.append(full[:, :kl].contiguous())
    w6 = torch.zeros(shapes[6], dtype=torch.float32, device=dev)                # only row 0 of lin6 reaches the loss
    w6[0] = (ybar[:, None] * in_v[6] + in_d[6]).sum(dim=0)[:156]
    b6 = torch.zeros(129, dtype=torch.float32, device=dev)
    b6[0] = ybar.sum()
    dW.append(w6)
    db.append(b6)
    return {"weight": dW, "bias": db, "volumes": dvols}


def sdf_smooth_backward(pts, sbar, volumes, packed, want_dvols=True, dvols=None):
    """Gradients of sum_n sbar_n . smooth_n (smooth = H.1 of the SDF, sdf_network.py:143-150) w.r.t. the EFFECTIVE matrices /
    biases of lin0..lin6 and the sparse feature rows (surf_sdf_smooth_backward; the batch reductions are surf_colgram).
    Returns the same dictionary as sdf_backward.  dvols: the (N_s, 8) gradient rows of an earlier sdf_backward to ACCUMULATE into
    (the kernel adds with atomics either way: saves a zero fill and a sum over ~10 M rows)."""
    _chk(pts, torch.float32, "pts")
    _chk(sbar, torch.float32, "sbar")
    _chk(packed, torch.float32, "packed weights")
    n, dev = pts.shape[0], pts.device
    xin = torch.empty(7, 4, n, 160, dtype=torch.float32, device=dev)
    ab = torch.empty(6, 4, n, 128, dtype=torch.float32, device=dev)
    if dvols is not None:
        assert len(dvols) == volumes.n and all(d.shape == v.shape and d.is_contiguous() for d, v in zip(dvols, volumes.vols))
    else:
        dvols = [torch.zeros_like(v) for v in volumes.vols] if want_dvols else None
    with _timed("sdf_smooth_bwd", n):
        rc = _lib.lib().surf_sdf_smooth_backward(_p(pts), _p(sbar), n, volumes._vp, volumes._tp, volumes._dp, volumes.n,
                                                 _ptr_array(dvols) if dvols is not None else None, _p(packed), _p(xin), _p(ab), _stream())
    _lib.check(rc, "surf_sdf_smooth_backward")
    shapes = [(128, 27), (128, 156), (101, 156), (128, 156), (128, 156), (128, 156), (129, 156)]
    dW, db = [], []
    for l in range(6):
        nl, kl = shapes[l]
        dW.append(colgram(ab[l].reshape(4 * n, 128)[:, :nl], xin[l].reshape(4 * n, 160)[:, :kl]))   # the four streams in one GEMM
        db.append(ab[l][0].sum(dim=0)[:nl].contiguous())
    w6 = torch.zeros(shapes[6], dtype=torch.float32, device=dev)                     # S = lin6 row 0 . (mixed input of lin6)
    w6[0] = xin[6][3].sum(dim=0)[:156]
    dW.append(w6)
    db.append(torch.zeros(129, dtype=torch.float32, device=dev))
    return {"weight": dW, "bias": db, "volumes": dvols}


# precision of the weight-gradient reductions (surf_colgram_p): 0 = fp32-equivalent, 1 = operands rounded to bf16 (fp32
# accumulate).  Set through `set_train_precision` (conf key `train_precision` of the model / bench --train-precision).
TRAIN_PRECISIONS = {"fp32": 0, "bf16": 1}
colgram_precision = 0
# thin sparse convolutions (a channel count of 8) on the matrix cores (csrc/spconv_mfma.hip, spconv_thin_mfma_kernel): "none"
# (default: the per-voxel FMA kernels of spconv.hip - both forms wait for the 27 row gathers of a site and the FMA form hides them
# better: profiles/r06_train_experiments.txt), "bf16" = under the bf16 training policy, "all" = also fp32-equivalent.  A/B switch.
thin_mfma = os.environ.get("SURF_THIN_MFMA", "none")
# bf16 ROW STORAGE of the sparse U-Net under the bf16 training policy (round 6): activations with 16 channels get a bf16 shadow
# (written by the BatchNorm apply / backward kernels in the same pass) that the (16 -> 8) thin convolutions gather from - the one
# channel pair where halving the row bytes pays (scripts/time_spconv_rows16.py: -35 .. -45 % per launch).  "0" = off (A/B switch).
layout_cache = os.environ.get("SURF_LAYOUT_CACHE", "1") != "0"   # FPN weight layouts cached per parameter version + the U-Nets' dgrad kernels built in the forward (A/B)
bf16_rows = os.environ.get("SURF_BF16_ROWS", "1") != "0"
bf16_rows_all_modes = os.environ.get("SURF_BF16_ROWS") == "all"      # also the stride-2 / transposed (16 -> 8) layers (measured: no gain)


def rows_to_bf16(x):
    """(n, C) fp32 rows -> their bf16 (round-to-nearest-even) bits as an (n, C) int16 tensor."""
    _chk(x, torch.float32, "x")
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    if x.numel():
        _lib.check(_lib.lib().surf_rows_to_bf16(_p(x), x.numel(), _p(out), _stream()), "surf_rows_to_bf16")
    return out


def set_train_precision(name):
    """Reduced-precision policy of the TRAINING step (BASELINE configs[3] "bf16").  "fp32" (default):
    every backward reduction fp32-equivalent.  "bf16": the operands of the weight-gradient reductions dW = adj^T in (the
    per-sample adjoint and input rows written by surf_sdf_backward / surf_sdf_smooth_backward / surf_blend_backward and the
    sparse U-Net's out_lin) are rounded to bf16 on load and multiplied on the matrix cores with fp32 accumulation.  Master
    weights, optimiser state, the adjoint propagation of the MLPs and every gather / scatter kernel stay fp32.  Round 5: the
    wide layers of the sparse U-Net (both channel counts >= 16, the matrix-core kernel) run their TRAINING forward and their
    input gradients on bf16-rounded operands too - one product per k-step instead of the exact split's six - which is what
    autocast(bf16) does to a convolution; the thin layers (gather bound) and the inference path are untouched."""
    global colgram_precision
    if name not in TRAIN_PRECISIONS:
        raise ValueError(f"train precision must be one of {sorted(TRAIN_PRECISIONS)}, got {name!r}")
    colgram_precision = TRAIN_PRECISIONS[name]
    return name


class precision_scope:
    """`with ops.precision_scope(p):` - run a BACKWARD under the training-precision policy its forward was recorded with
    (p = the value of `colgram_precision` then), whatever another model's forward has set since; restores the previous one."""

    def __init__(self, precision):
        self.precision = precision

    def __enter__(self):
        global colgram_precision
        self.saved = colgram_precision
        if self.precision is not None:
            colgram_precision = int(self.precision)

    def __exit__(self, *exc):
        global colgram_precision
        colgram_precision = self.saved
        return False


def colgram(A, X, with_sum=False, out=None, precision=None):
    """out (M, N [+1]) = A^T [X | 1] for row-major 2-D views A (rows, M), X (rows, N) that share contiguous rows (column
    slices of a wider buffer are fine: the row stride is taken from the view).  `out` given: accumulated into.
    precision: None = the module's training policy (`set_train_precision`), 0 = fp32-equivalent, 1 = bf16 operands."""
    assert A.dim() == 2 and X.dim() == 2 and A.shape[0] == X.shape[0] and A.stride(1) == 1 and X.stride(1) == 1
    assert A.dtype == torch.float32 and X.dtype == torch.float32 and A.is_cuda and X.is_cuda
    rows, M, N = int(A.shape[0]), int(A.shape[1]), int(X.shape[1])
    acc = out is not None
    if out is None:
        out = torch.empty(M, N + (1 if with_sum else 0), dtype=torch.float32, device=A.device)
    if rows == 0:
        return out if acc else out.zero_()
    ws = torch.empty(_lib.lib().surf_colgram_workspace_floats(rows, M, N), dtype=torch.float32, device=A.device)
    with _timed("colgram", rows * M * (N + (1 if with_sum else 0))):
        rc = _lib.lib().surf_colgram_p(_p(A), int(A.stride(0)), M, _p(X), int(X.stride(0)), N, rows, int(with_sum), int(acc),
                                       int(colgram_precision if precision is None else precision), _p(ws), _p(out), _stream())
    _lib.check(rc, "surf_colgram_p")
    return out


_BLEND_LAYERS = [  # (state_dict prefix, in, out, IN column, ADJ column) of surf_blend_backward's rows
    ("ray_dir_fc.0", 4, 16, 0, 4), ("ray_dir_fc.2", 16, 19, 20, 36), ("base_fc.0", 57, 64, 55, 112), ("base_fc.2", 64, 32, 176, 240),
    ("vis_fc.0", 32, 32, 272, 304), ("vis_fc.2", 32, 33, 336, 368), ("vis_fc2.0", 32, 32, 401, 433), ("vis_fc2.2", 32, 1, 465, 497),
    ("rgb_fc.0", 37, 16, 498, 535), ("rgb_fc.2", 16, 8, 551, 567), ("rgb_fc.4", 8, 1, 575, 583)]


def blend_backward(pts, active_idx, gcolor, feats_t4, imgs_t4, cams, raw_weights, want_color=False, gfeats_t4=None):
    """Gradients of sum_n gcolor_n . colour_n w.r.t. the blending network's parameters (surf_blend_backward; the batch
    reductions are surf_colgram_p: matrix cores).  raw_weights: device tensor of blend_raw_weights(sd) (packing.blend_raw_device).
    gfeats_t4: four texel4 maps shaped like feats_t4 that accumulate the gradient of the sampled feature channels.
    Returns {state_dict name: gradient} (+ "_color": the recomputed colours of the active samples, if asked)."""
    _chk(pts, torch.float32, "pts")
    _chk(gcolor, torch.float32, "gcolor")
    _chk(raw_weights, torch.float32, "raw weights")
    assert len(feats_t4) == 4 and gcolor.shape[0] == pts.shape[0]
    dev = pts.device
    n = int(active_idx.shape[0]) if active_idx is not None else pts.shape[0]
    V = cams.nv - 1
    ROW = _lib.lib().surf_blend_backward_row_floats()
    rows = torch.empty(max(n, 1), V, ROW, dtype=torch.float32, device=dev)
    ds = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
    color = torch.empty(max(n, 1), 3, dtype=torch.float32, device=dev) if want_color else None
    out = {}
    if n > 0:
        hw = (ctypes.c_int * 8)(*[int(v) for f in feats_t4 for v in f.shape[1:3]])
        intr16 = np.ascontiguousarray(cams.intrs.reshape(cams.nv, -1))
        with _timed("blend_bwd", n * V):
            rc = _lib.lib().surf_blend_backward(_p(pts), _p(active_idx), n, _p(gcolor), _ptr_array(list(feats_t4)), hw, _p(imgs_t4),
                                                cams.nv, _np_ptr(intr16), _np_ptr(cams.w2c), _np_ptr(cams.c2w), _p(raw_weights),
                                                _p(rows), _p(ds), _p(color),
                                                None if gfeats_t4 is None else _ptr_array(list(gfeats_t4)), _stream())
        _lib.check(rc, "surf_blend_backward")
    flat = rows[:n].reshape(-1, ROW)
    for name, cin, cout, c_in, c_ad in _BLEND_LAYERS:
        g = colgram(flat[:, c_ad:c_ad + cout], flat[:, c_in:c_in + cin], with_sum=True)
        out[name + ".weight"] = g[:, :cin].contiguous()
        out[name + ".bias"] = g[:, cin].contiguous()
    out["s"] = ds[:n].sum().reshape(())
    if want_color:
        out["_color"] = color[:n]
    return out


class Cameras:
    """Host copies of the 4x4 camera matrices the kernels take by value."""

    def __init__(self, intrs, c2ws):
        self.nv = int(intrs.shape[0])
        c2w_cpu = c2ws.detach().to("cpu", torch.float32)
        self.intrs = np.ascontiguousarray(intrs.detach().to("cpu", torch.float32).numpy())
        self.c2w = np.ascontiguousarray(c2w_cpu.numpy())
        self.w2c = np.ascontiguousarray(torch.inverse(c2w_cpu).numpy())
        self.rot_ref = np.ascontiguousarray(torch.inverse(c2w_cpu[0, :3, :3]).numpy())


def blend(pts, feats_t4, imgs_t4, cams, packed, mask=None, compact_active=True, active_idx=None, active_count=None):
    """projector.py:501-556 + blending_network.py:69-118.  feats_t4: list fine -> coarse of (nv,H,W,4).
    Returns (color (n,3), n_valid (n) uint8); masked-out rows are zero."""
    _chk(pts, torch.float32, "pts")
    n = pts.shape[0]
    dev = pts.device
    for f in feats_t4:
        _chk(f, torch.float32, "feature map")
    _chk(imgs_t4, torch.float32, "imgs")
    color = torch.zeros(n, 3, dtype=torch.float32, device=dev)
    nvalid = torch.zeros(n, dtype=torch.uint8, device=dev)
    hw = (ctypes.c_int * (2 * len(feats_t4)))(*[int(v) for f in feats_t4 for v in f.shape[1:3]])
    fp = _ptr_array(feats_t4)
    idx, n_eval = None, n
    if active_idx is not None:
        idx, n_eval = active_idx, int(active_idx.shape[0])
    elif mask is not None and compact_active:
        idx = compact(mask)
        n_eval = int(idx.shape[0])
    if n_eval == 0:
        return color, nvalid
    precision = blend_packed_precision(packed)     # which kernel the packed weights were laid out for
    if precision == "f32" and active_count is not None:
        raise ValueError("active_count needs one of the split blend kernels")
    if precision == "f32":
        rc = _lib.lib().surf_blend(_p(pts), _p(None if idx is not None else mask), _p(idx), n_eval, fp, hw, len(feats_t4),
                                   _p(imgs_t4), cams.nv, _np_ptr(cams.intrs), _np_ptr(cams.w2c), _np_ptr(cams.c2w), _p(packed),
                                   _p(color), _p(nvalid), _stream())
    else:
        need = _lib.lib().surf_blend_split_scratch_bytes(int(n_eval), cams.nv)
        key = ("blend", dev.index if dev.index is not None else torch.cuda.current_device())
        scratch = _scratch_cache.get(key)
        if scratch is None or scratch.numel() < need:
            scratch = torch.empty(need, dtype=torch.uint8, device=dev)
            _scratch_cache[key] = scratch
        if active_count is not None:           # the count stays on the device (compact_counted)
            _chk(active_count, torch.int32, "active_count")
            rc = _lib.lib().surf_blend_split_dn(_p(pts), _p(idx), n_eval, _p(active_count), fp, hw, len(feats_t4), _p(imgs_t4),
                                                cams.nv, _np_ptr(cams.intrs), _np_ptr(cams.w2c), _np_ptr(cams.c2w), _p(packed),
                                                _BLEND_ID[precision], _p(color), _p(nvalid), _p(scratch), _stream())
        else:
            rc = _lib.lib().surf_blend_split(_p(pts), _p(None if idx is not None else mask), _p(idx), n_eval, fp, hw, len(feats_t4),
                                             _p(imgs_t4), cams.nv, _np_ptr(cams.intrs), _np_ptr(cams.w2c), _np_ptr(cams.c2w),
                                             _p(packed), _BLEND_ID[precision], _p(color), _p(nvalid), _p(scratch), _stream())
    _lib.check(rc, "surf_blend" if precision == "f32" else f"surf_blend_split({precision})")
    return color, nvalid


def composite(sdf, grad, color, n_valid, setup, rays_d, inv_s, cos_anneal_ratio, cams, per_sample=True, want_z0=False):
    """implicit_surface.py:126-166,181-216.  Returns the per-ray dict (+ weights / inside_sphere if per_sample;
    + z_sdf0, the zero crossing's ray parameter, if want_z0)."""
    R, S = setup["mid_z"].shape
    dev = sdf.device
    f32 = dict(dtype=torch.float32, device=dev)
    out = {
        "color_fine": torch.empty(R, 3, **f32), "render_depth": torch.empty(R, **f32),
        "sdf_depth": torch.empty(R, 1, **f32), "normal": torch.empty(R, 3, **f32),
        "normal_val": torch.empty(R, 3, **f32), "valid_mask": torch.empty(R, 1, dtype=torch.uint8, device=dev),
        "mid_inside_sphere": torch.empty(R, 1, dtype=torch.uint8, device=dev), "eik": torch.empty(R, 2, **f32),
    }
    if per_sample:
        out["weights"] = torch.empty(R, S, **f32)
        out["inside_sphere"] = torch.empty(R, S, **f32)
    if want_z0:
        out["z_sdf0"] = torch.empty(R, **f32)
    rc = _lib.lib().surf_composite(_p(sdf), _p(grad), _p(color), _p(n_valid), _p(setup["mid_z"]), _p(setup["dists"]),
                                   _p(setup["pts"]), _p(setup["vmask"]), _p(rays_d), R, S, ctypes.c_float(inv_s),
                                   ctypes.c_float(cos_anneal_ratio), _np_ptr(cams.rot_ref), _p(out["color_fine"]),
                                   _p(out["render_depth"]), _p(out["sdf_depth"]), _p(out["normal"]),
                                   _p(out["normal_val"]), _p(out["valid_mask"]), _p(out["mid_inside_sphere"]),
                                   _p(out.get("weights")), _p(out.get("inside_sphere")), _p(out["eik"]), _p(out.get("z_sdf0")),
                                   _stream())
    _lib.check(rc, "surf_composite")
    return out


def composite_backward(sdf, grad, color, setup, rays_d, inv_s, cos_anneal_ratio, cams, g_color, g_depth=None, eik_scale=0.0,
                       eik_upstream=None):
    """Backward of `composite` w.r.t. (sdf, grad, color, inv_s) for upstream gradients of colour_fine (R,3) and
    render_depth (R) and the eikonal term (eik_scale = dL/d gradient_error / (sum relax + 1e-5)).
    Returns (d_sdf (R*S,), d_grad (R*S,3), d_color (R*S,3), d_inv_s scalar tensor)."""
    R, S = setup["mid_z"].shape
    dev = sdf.device
    _chk(g_color, torch.float32, "g_color")
    d_sdf = torch.empty(R * S, dtype=torch.float32, device=dev)
    d_grad = torch.empty(R * S, 3, dtype=torch.float32, device=dev)
    d_color = torch.empty(R * S, 3, dtype=torch.float32, device=dev)
    d_is = torch.empty(R, dtype=torch.float32, device=dev)
    if eik_upstream is not None:        # dL/d gradient_error as a device scalar: multiplied in inside the kernel, never read back
        eik_upstream = _chk(eik_upstream.detach().reshape(1).float().contiguous(), torch.float32, "eik_upstream")
    rc = _lib.lib().surf_composite_backward_s(_p(sdf), _p(grad), _p(color), _p(setup["mid_z"]), _p(setup["dists"]), _p(setup["pts"]),
                                              _p(setup["vmask"]), _p(rays_d), R, S, ctypes.c_float(inv_s),
                                              ctypes.c_float(cos_anneal_ratio), _np_ptr(cams.rot_ref), _p(g_color), _p(g_depth),
                                              ctypes.c_float(float(eik_scale)), _p(eik_upstream), _p(d_sdf), _p(d_grad), _p(d_color),
                                              _p(d_is), _stream())
    _lib.check(rc, "surf_composite_backward_s")
    return d_sdf, d_grad, d_color, d_is.sum(dtype=torch.float64).float()


def upsample_bilinear_t4(x, H, W):
    """F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False) on a texel4 map (n,h,w,4)."""
    _chk(x, torch.float32, "x")
    n, h, w, c = x.shape
    assert c == 4
    out = torch.empty(n, H, W, 4, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().surf_upsample_bilinear_t4(_p(x), n, h, w, int(H), int(W), _p(out), _stream()), "surf_upsample_bilinear_t4")
    return out


def surface_points(rays_o, rays_d, z_sdf0, z_vals):
    """implicit_surface.py:217-220: the zero crossing's ray parameter zeroed outside [0, max(z_vals)], then o + d z."""
    _chk(rays_o, torch.float32, "rays_o")
    _chk(rays_d, torch.float32, "rays_d")
    _chk(z_sdf0, torch.float32, "z_sdf0")
    _chk(z_vals, torch.float32, "z_vals")
    R = rays_o.shape[0]
    ws = torch.empty(1, dtype=torch.int32, device=rays_o.device)
    pts = torch.empty(R, 3, dtype=torch.float32, device=rays_o.device)
    _lib.check(_lib.lib().surf_surface_points(_p(rays_o), _p(rays_d), _p(z_sdf0), R, _p(z_vals), z_vals.numel(), _p(ws), _p(pts),
                                              _stream()), "surf_surface_points")
    return pts


def patch_warp(pts, grads, maps_t4, cams, patch_size=11):
    """projector.py:560-645 (+ the normal's normalisation and rotation of implicit_surface.py:224-228).
    maps_t4: the three finest feature levels as texel4 (nv,H,W,4) at full resolution; cams: ops.Cameras.
    Returns (ref_gray_val (1,R,p*p,12), sampled_gray_val (nv-1,R,p*p,12))."""
    _chk(pts, torch.float32, "pts")
    _chk(grads, torch.float32, "grads")
    for m in maps_t4:
        _chk(m, torch.float32, "feature map")
    nv, H, W, _ = maps_t4[0].shape
    assert len(maps_t4) == 3 and all(tuple(m.shape) == (nv, H, W, 4) for m in maps_t4)
    R = pts.shape[0]
    dev = pts.device
    if not hasattr(cams, "kinv_ref"):
        cams.kinv_ref = np.ascontiguousarray(torch.inverse(torch.from_numpy(cams.intrs))[0, :3, :3].contiguous().numpy())
    K, kinv, c2w = cams.intrs, cams.kinv_ref, cams.c2w
    assert cams.nv == nv
    P = patch_size * patch_size
    ref = torch.empty(1, R, P, 12, dtype=torch.float32, device=dev)
    src = torch.empty(nv - 1, R, P, 12, dtype=torch.float32, device=dev)
    rc = _lib.lib().surf_patch_warp(_p(pts), _p(grads), R, _ptr_array(list(maps_t4)), nv, H, W, _np_ptr(K), _np_ptr(kinv),
                                    _np_ptr(c2w), int(patch_size), _p(ref), _p(src), _stream())
    _lib.check(rc, "surf_patch_warp")
    return ref, src


def photometric_loss(depth, imgs_t4, mask_ref, cams, ref_idx=0, topk=2, return_warp=False, return_state=False):
    """compute_ptloss (losses/photometric_loss.py:54-125) of one (H,W) depth map of view ref_idx: scalar tensor.
    imgs_t4 (nv,H,W,4) texel4; cams: ops.Cameras (host matrices).  return_state: (loss, (warp, column sums)) - what
    photometric_loss_backward would otherwise recompute with a second run of the forward kernel."""
    _chk(depth, torch.float32, "depth")
    _chk(imgs_t4, torch.float32, "imgs_t4")
    _chk(mask_ref, torch.float32, "mask_ref")
    nv, H, W, _ = imgs_t4.shape
    assert tuple(depth.shape) == (H, W) and tuple(mask_ref.shape) == (H, W) and cams.nv == nv
    dev = depth.device
    warp = torch.empty(nv - 1, H, W, 4, dtype=torch.float32, device=dev)
    terms = torch.empty(H, W, 8, dtype=torch.float32, device=dev)
    intr16 = np.ascontiguousarray(cams.intrs.reshape(nv, -1))
    assert intr16.shape[1] == 16
    rc = _lib.lib().surf_ptloss_terms(_p(imgs_t4), nv, H, W, _p(depth), _p(mask_ref), int(ref_idx), int(topk), _np_ptr(intr16),
                                      _np_ptr(cams.c2w), _np_ptr(cams.w2c), _p(warp), _p(terms), _stream())
    _lib.check(rc, "surf_ptloss_terms")
    t = terms.view(-1, 8).sum(dim=0, dtype=torch.float64)       # columns [l1 m, gx mx, gy my, ssim m | m, mx, my, m]
    loss = (t[:4] / (t[4:] + 1e-8)).sum().float()
    if return_state:
        return loss, (warp, t)
    return (loss, warp) if return_warp else loss


def photometric_loss_backward(depth, imgs_t4, mask_ref, cams, ref_idx=0, topk=2, upstream=1.0, state=None):
    """d (upstream * photometric_loss(depth, ...)) / d depth (H,W) (surf_ptloss_backward).  state: the forward's (warp, column
    sums) (photometric_loss(..., return_state=True)); None: the forward kernel is re-run for them.  upstream: float or 0-d
    device tensor."""
    _chk(depth, torch.float32, "depth")
    nv, H, W, _ = imgs_t4.shape
    dev = depth.device
    intr16 = np.ascontiguousarray(cams.intrs.reshape(nv, -1))
    if state is not None:
        warp, t = state
    else:
        warp = torch.empty(nv - 1, H, W, 4, dtype=torch.float32, device=dev)
        terms = torch.empty(H, W, 8, dtype=torch.float32, device=dev)
        rc = _lib.lib().surf_ptloss_terms(_p(imgs_t4), nv, H, W, _p(depth), _p(mask_ref), int(ref_idx), int(topk), _np_ptr(intr16),
                                          _np_ptr(cams.c2w), _np_ptr(cams.w2c), _p(warp), _p(terms), _stream())
        _lib.check(rc, "surf_ptloss_terms")
        t = terms.view(-1, 8).sum(dim=0, dtype=torch.float64)
    coef = (upstream / (t[4:] + 1e-8)).float().contiguous()
    g_warp = torch.empty_like(warp)
    g_depth = torch.empty(H, W, dtype=torch.float32, device=dev)
    rc = _lib.lib().surf_ptloss_backward(_p(imgs_t4), nv, H, W, _p(depth), _p(mask_ref), int(ref_idx), int(topk), _np_ptr(intr16),
                                         _np_ptr(cams.c2w), _np_ptr(cams.w2c), _p(warp), _p(coef), _p(g_warp), _p(g_depth), _stream())
    _lib.check(rc, "surf_ptloss_backward")
    return g_depth


def patch_warp_tangent(pts, dirs, grads, maps_t4, cams, patch_size=11):
    """patch_warp + the derivatives of both patch stacks along d pts / d z0 = dirs (R,3) (surf_patch_warp_tangent).
    Returns (ref, src, ref_tan, src_tan)."""
    for t, nm in ((pts, "pts"), (dirs, "dirs"), (grads, "grads")):
        _chk(t, torch.float32, nm)
    nv, H, W, _ = maps_t4[0].shape
    R, dev, P = pts.shape[0], pts.device, patch_size * patch_size
    if not hasattr(cams, "kinv_ref"):
        cams.kinv_ref = np.ascontiguousarray(torch.inverse(torch.from_numpy(cams.intrs))[0, :3, :3].contiguous().numpy())
    ref, ref_t = (torch.empty(1, R, P, 12, dtype=torch.float32, device=dev) for _ in range(2))
    src, src_t = (torch.empty(nv - 1, R, P, 12, dtype=torch.float32, device=dev) for _ in range(2))
    rc = _lib.lib().surf_patch_warp_tangent(_p(pts), _p(dirs), _p(grads), R, _ptr_array(list(maps_t4)), nv, H, W, _np_ptr(cams.intrs),
                                            _np_ptr(cams.kinv_ref), _np_ptr(cams.c2w), int(patch_size), _p(ref), _p(src), _p(ref_t),
                                            _p(src_t), _stream())
    _lib.check(rc, "surf_patch_warp_tangent")
    return ref, src, ref_t, src_t


def lncc_jvp(ref, src, ref_tan, src_tan):
    """(ncc (R,1), d ncc / d z0 (R,)) of compute_LNCC2 for patch tangents (surf_lncc_jvp)."""
    nsrc, R, P, C = src.shape
    ncc = torch.empty(R, 1, dtype=torch.float32, device=ref.device)
    d = torch.empty(R, dtype=torch.float32, device=ref.device)
    rc = _lib.lib().surf_lncc_jvp(_p(ref), _p(src), _p(ref_tan), _p(src_tan), R, int(nsrc), int(P), int(C), _p(ncc), _p(d), _stream())
    _lib.check(rc, "surf_lncc_jvp")
    return ncc, d


def crossing_backward(sdf, vmask, mid_z, zmax, g_z0, d_sdf):
    """d_sdf (R*S,) += g_z0 (R,) d z0 / d sdf at the first zero crossing of every ray (surf_crossing_backward); zmax: 0-d tensor."""
    R, S = mid_z.shape
    _chk(d_sdf, torch.float32, "d_sdf")
    rc = _lib.lib().surf_crossing_backward(_p(sdf), _p(vmask), _p(mid_z), R, S, _p(zmax.reshape(1).float().contiguous()),
                                           _p(g_z0.float().contiguous()), _p(d_sdf), _stream())
    _lib.check(rc, "surf_crossing_backward")
    return d_sdf


def lncc(ref_gray_val, sampled_gray_val):
    """compute_LNCC2 (losses/ncc.py:7-51): ref (1,R,P,C), src (nsrc,R,P,C) -> (R,1)."""
    _chk(ref_gray_val, torch.float32, "ref_gray_val")
    _chk(sampled_gray_val, torch.float32, "sampled_gray_val")
    nsrc, R, P, C = sampled_gray_val.shape
    assert tuple(ref_gray_val.shape) == (1, R, P, C)
    out = torch.empty(R, 1, dtype=torch.float32, device=ref_gray_val.device)
    if R > 0:
        _lib.check(_lib.lib().surf_lncc(_p(ref_gray_val), _p(sampled_gray_val), R, int(nsrc), int(P), int(C), _p(out), _stream()),
                   "surf_lncc")
    return out


def lncc_backward(ref_gray_val, sampled_gray_val, g_ncc):
    """(d/d ref_gray_val, d/d sampled_gray_val) of sum_r g_ncc[r] * compute_LNCC2(ref, src)[r] (surf_lncc_backward)."""
    _chk(ref_gray_val, torch.float32, "ref_gray_val")
    _chk(sampled_gray_val, torch.float32, "sampled_gray_val")
    nsrc, R, P, C = sampled_gray_val.shape
    g = _chk(g_ncc.reshape(-1).float().contiguous(), torch.float32, "g_ncc")
    assert g.shape[0] == R
    g_ref, g_src = torch.empty_like(ref_gray_val), torch.empty_like(sampled_gray_val)
    if R > 0:
        _lib.check(_lib.lib().surf_lncc_backward(_p(ref_gray_val), _p(sampled_gray_val), _p(g), R, int(nsrc), int(P), int(C),
                                                 _p(g_ref), _p(g_src), _stream()), "surf_lncc_backward")
    return g_ref, g_src


# ------------------------------------------------------------------------------------------------
# volume build (surf.py:80-131)
# ------------------------------------------------------------------------------------------------


def _cams_ext(cams, intrs, c2ws):
    """Extra host matrices of the matching field: inverse(intrs)[:, :3, :3], inverse(c2w[:, :3, :3])."""
    if not hasattr(cams, "kinv"):
        i_cpu = intrs.detach().to("cpu", torch.float32)
        c_cpu = c2ws.detach().to("cpu", torch.float32)
        cams.kinv = np.ascontiguousarray(torch.inverse(i_cpu)[:, :3, :3].contiguous().numpy())
        cams.rinv = np.ascontiguousarray(torch.inverse(c_cpu[:, :3, :3]).contiguous().numpy())
    return cams


def upsample_filter(parents, D, depths, cams, depth_range):
    """volume.py:35-52 + 134-165: flags (8 N,) uint8 over the children of `parents` (N,3) int32."""
    _chk(parents, torch.int32, "parents")
    _chk(depths, torch.float32, "depths")
    nv, H, W = depths.shape
    flags = torch.empty(parents.shape[0] * 8, dtype=torch.uint8, device=parents.device)
    rc = _lib.lib().surf_upsample_filter(_p(parents), parents.shape[0], int(D), _p(depths), nv, H, W, _np_ptr(cams.intrs),
                                         _np_ptr(cams.w2c), ctypes.c_float(float(depth_range)), _p(flags), _stream())
    _lib.check(rc, "surf_upsample_filter")
    return flags


def agg_mlp_host(sd, prefix="volume."):
    return np.concatenate([_host_f32(sd[prefix + k]).reshape(-1) for k in
                           ("agg_mlp.0.weight", "agg_mlp.0.bias", "agg_mlp.2.weight", "agg_mlp.2.bias")])


def costvol(feats_t4_c2f, stage, D, cams, agg, parents=None, idx=None):
    """volume.py:54-97 on the full lattice (parents/idx None) or on children idx of parents.
    feats_t4_c2f: texel4 pyramids coarse -> fine.  Returns coords (n,3) int32, feat (n,8), keep (n,) uint8."""
    dev = feats_t4_c2f[0].device
    n = int(D) ** 3 if idx is None else int(idx.shape[0])
    coords = torch.empty(n, 3, dtype=torch.int32, device=dev)
    feat = torch.empty(n, 8, dtype=torch.float32, device=dev)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    hw = (ctypes.c_int * 8)(*[int(v) for f in feats_t4_c2f for v in f.shape[1:3]])
    fp = _ptr_array(feats_t4_c2f)
    agg = np.ascontiguousarray(agg, dtype=np.float32)
    rc = _lib.lib().surf_costvol(_p(parents), _p(idx), n, int(D), fp, hw, int(stage), cams.nv, _np_ptr(cams.intrs),
                                 _np_ptr(cams.w2c), _np_ptr(agg), _p(coords), _p(feat), _p(keep), _stream())
    _lib.check(rc, "surf_costvol")
    return coords, feat, keep


def compact(flags):
    """Ascending indices of the set flags (int32).  One host sync to size the result -- the reference syncs at the
    same places (boolean-mask indexing, volume.py:165-166, surf.py:104-108)."""
    _chk(flags, torch.uint8, "flags")
    n = flags.numel()
    dev = flags.device
    ws = torch.empty(_lib.lib().surf_compact_workspace_ints(n), dtype=torch.int32, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().surf_compact(_p(flags), n, _p(ws), _p(idx), _p(total), _stream()), "surf_compact")
    return idx[:int(total.item())]


def compact_counted(flags):
    """compact without the host round trip: (idx (n,) int32 of which the first `count` entries are the ascending indices of
    the set flags, count (1,) int32 ON THE DEVICE).  For consumers that read the count from device memory (sdf_mlp / blend
    with `active_count`)."""
    _chk(flags, torch.uint8, "flags")
    n = flags.numel()
    dev = flags.device
    ws = torch.empty(_lib.lib().surf_compact_workspace_ints(n), dtype=torch.int32, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().surf_compact(_p(flags), n, _p(ws), _p(idx), _p(total), _stream()), "surf_compact")
    return idx, total


def gather_rows(src, idx, shift=0, dst=None, dst_off=0):
    """dst[i, off:off+w] = src[idx[i] >> shift]; src rows of 32-bit elements."""
    assert src.is_contiguous() and src.element_size() == 4 and src.is_cuda
    _chk(idx, torch.int32, "idx")
    w = src.shape[1]
    n = idx.shape[0]
    if dst is None:
        dst = torch.empty(n, w, dtype=src.dtype, device=src.device)
    rc = _lib.lib().surf_gather_rows(_p(src), _p(idx), n, w, int(shift), dst.shape[1], int(dst_off), _p(dst), _stream())
    _lib.check(rc, "surf_gather_rows")
    return dst


def compose_index(a, b):
    out = torch.empty_like(b)
    _lib.check(_lib.lib().surf_compose_index(_p(a), _p(b), b.shape[0], _p(out), _stream()), "surf_compose_index")
    return out


def densify(coords, rows, D, prev=None):
    """volume.py:99-132: dense matching volume (D,D,D) f32 and index table (D,D,D) int32; logit = rows[:, 0]."""
    _chk(coords, torch.int32, "coords")
    _chk(rows, torch.float32, "rows")
    dev = coords.device
    dense = torch.empty(D, D, D, dtype=torch.float32, device=dev)
    table = torch.empty(D, D, D, dtype=torch.int32, device=dev)
    rc = _lib.lib().surf_densify(_p(coords), _p(rows), rows.shape[1], coords.shape[0], int(D), _p(prev), _p(dense), _p(table),
                                 _stream())
    _lib.check(rc, "surf_densify")
    return dense, table


def matching_depth(mvol, cams, near_fars, H, W, res_level, n, pre_depths=None, ratio_cur=1.0, ratio_prev=1.0, return_lr=False,
                   jitter=None, saved=None):
    """matching_field.py:73-141.  Returns depth maps (nv,H,W) (and, with return_lr, the (nv,h,w) maps rendered at the
    reduced resolution before the bilinear upsample).  jitter (nv, h*w, 2): the train-mode per-ray, per-band z shifts.
    saved: a dict that receives "stats" (nv,h,w,4) = the per-ray softmax statistics matching_depth_backward can start from."""
    _chk(mvol, torch.float32, "matching volume")
    dev = mvol.device
    h, w = H // res_level, W // res_level
    lin_x = _linspace_dev(0, W - 1, w, dev)
    lin_y = _linspace_dev(0, H - 1, h, dev)
    lin_n = _linspace_dev(0.0, 1.0, n, dev)
    nf = _near_fars_host(near_fars)
    lr = torch.empty(cams.nv, h, w, dtype=torch.float32, device=dev)
    full = torch.empty(cams.nv, H, W, dtype=torch.float32, device=dev)
    if jitter is not None:
        _chk(jitter, torch.float32, "jitter")
        assert tuple(jitter.shape) == (cams.nv, h * w, 2)
    stats = torch.empty(cams.nv, h, w, 4, dtype=torch.float32, device=dev) if saved is not None else None
    rc = _lib.lib().surf_matching_depth(_p(mvol), int(mvol.shape[0]), cams.nv, _np_ptr(cams.kinv), _np_ptr(cams.c2w),
                                        _np_ptr(cams.rinv), _np_ptr(nf), H, W, h, w, _p(lin_x), _p(lin_y), _p(lin_n), int(n),
                                        _p(pre_depths), ctypes.c_float(float(ratio_cur)), ctypes.c_float(float(ratio_prev)),
                                        _p(jitter), _p(lr), _p(full), _p(stats), _stream())
    if saved is not None:
        saved["stats"] = stats
    _lib.check(rc, "surf_matching_depth")
    return (full, lr) if return_lr else full


def matching_depth_backward(mvol, cams, near_fars, H, W, res_level, n, g_full, pre_depths=None, ratio_cur=1.0, ratio_prev=1.0,
                            jitter=None, dmvol=None, views=(0, 0), stats=None):
    """Backward of matching_depth w.r.t. the matching volume: g_full (nv,H,W) = d loss / d depth maps; only the maps of
    `views` = (reference view, src_idx) are read (the reference renders the others under no_grad).  Returns dmvol (D,D,D),
    accumulated into `dmvol` if given.  stats: the forward's per-ray softmax statistics (matching_depth(..., saved=)) for the SAME
    arguments and jitter - the kernel then walks the samples once instead of twice."""
    _chk(mvol, torch.float32, "matching volume")
    _chk(g_full, torch.float32, "g_full")
    dev = mvol.device
    h, w = H // res_level, W // res_level
    assert tuple(g_full.shape) == (cams.nv, H, W)
    lin_x = _linspace_dev(0, W - 1, w, dev)
    lin_y = _linspace_dev(0, H - 1, h, dev)
    lin_n = _linspace_dev(0.0, 1.0, n, dev)
    nf = _near_fars_host(near_fars)
    g_lr = torch.empty(cams.nv, h, w, dtype=torch.float32, device=dev)
    if stats is not None:
        _chk(stats, torch.float32, "stats")
        assert tuple(stats.shape) == (cams.nv, h, w, 4)
    if dmvol is None:
        dmvol = torch.zeros_like(mvol)
    n_views = len(set(int(v) for v in views))
    with _timed("matching_depth_bwd", n_views * h * w * int(n) * (1 if pre_depths is None else 2)):
        rc = _lib.lib().surf_matching_depth_backward(_p(mvol), int(mvol.shape[0]), cams.nv, _np_ptr(cams.kinv), _np_ptr(cams.c2w),
                                                     _np_ptr(cams.rinv), _np_ptr(nf), H, W, h, w, _p(lin_x), _p(lin_y), _p(lin_n),
                                                     int(n), _p(pre_depths), ctypes.c_float(float(ratio_cur)),
                                                     ctypes.c_float(float(ratio_prev)), _p(jitter), _p(g_full), int(views[0]), int(views[1]),
                                                     _p(g_lr), _p(dmvol), _p(stats), _stream())
    _lib.check(rc, "surf_matching_depth_backward")
    return dmvol


def densify_backward(coords, table, g_dense, g_rows, g_prev=None):
    """Backward of densify: g_rows[:, 0] += g_dense[coords]; g_prev (D/2)^3 (if given) accumulates the background's share."""
    _chk(coords, torch.int32, "coords")
    _chk(g_dense, torch.float32, "g_dense")
    _chk(g_rows, torch.float32, "g_rows")
    D = int(g_dense.shape[0])
    rc = _lib.lib().surf_densify_backward(_p(coords), coords.shape[0], D, _p(table), _p(g_dense), g_rows.shape[1], _p(g_rows),
                                          _p(g_prev), _stream())
    _lib.check(rc, "surf_densify_backward")
    return g_rows, g_prev


def scatter_rows_add(g_dst, idx, g_src, shift=0, dst_off=0):
    """Backward of gather_rows: g_src[idx[i] >> shift] += g_dst[i, off : off + w], w = g_src.shape[1]."""
    _chk(g_dst, torch.float32, "g_dst")
    _chk(g_src, torch.float32, "g_src")
    _chk(idx, torch.int32, "idx")
    rc = _lib.lib().surf_scatter_rows_add(_p(g_dst), _p(idx), idx.shape[0], g_src.shape[1], int(shift), g_dst.shape[1],
                                          int(dst_off), _p(g_src), _stream())
    _lib.check(rc, "surf_scatter_rows_add")
    return g_src


def costvol_backward(feats_t4_c2f, gfeats_t4_c2f, stage, D, cams, agg, coords, g, g_agg):
    """Backward of costvol for the kept voxels: accumulates into gfeats_t4_c2f[l] (l >= stage) and g_agg (49,)."""
    _chk(coords, torch.int32, "coords")
    _chk(g, torch.float32, "g")
    _chk(g_agg, torch.float32, "g_agg")
    assert g.shape[1] == 8 and g_agg.numel() == 49
    hw = (ctypes.c_int * 8)(*[int(v) for f in feats_t4_c2f for v in f.shape[1:3]])
    agg = np.ascontiguousarray(agg, dtype=np.float32)
    # 9 floats per (view, voxel) pair when the tile-sorted form runs (1.1 - 1.5 GB at the finest DTU stage), 4,096 floats otherwise
    ws = torch.empty(_lib.lib().surf_costvol_backward_workspace_floats_for(coords.shape[0], cams.nv, hw),
                     dtype=torch.float32, device=g.device)
    with _timed("costvol_bwd", int(coords.shape[0]) * cams.nv * (4 - int(stage))):
        rc = _lib.lib().surf_costvol_backward(_p(coords), _p(g), coords.shape[0], int(D), _ptr_array(feats_t4_c2f),
                                              _ptr_array(gfeats_t4_c2f), hw, int(stage), cams.nv, _np_ptr(cams.intrs),
                                              _np_ptr(cams.w2c), _np_ptr(agg), _p(ws), _p(g_agg), _stream())
    _lib.check(rc, "surf_costvol_backward")


def raster_first_hit(vertices, faces, intr, c2w, hw, upscale=1):
    """Face id of the first triangle hit by the ray through every sample of the (h*upscale, w*upscale) lattice
    torch.linspace(0, w-1, w*upscale) x torch.linspace(0, h-1, h*upscale) of one view; -1 where nothing is hit."""
    _chk(vertices, torch.float32, "vertices")
    _chk(faces, torch.int32, "faces")
    h, w = int(hw[0]), int(hw[1])
    Hup, Wup = int(h * upscale), int(w * upscale)
    K = np.ascontiguousarray(intr.detach().to("cpu", torch.float32)[:3, :3].contiguous().numpy())
    w2c = np.ascontiguousarray(torch.inverse(c2w.detach().to("cpu", torch.float32))[:3, :4].contiguous().numpy())
    zbuf = torch.full((Hup, Wup), -1, dtype=torch.int64, device=vertices.device)      # all ones = empty
    rc = _lib.lib().surf_raster_first_hit(_p(vertices), _p(faces), faces.shape[0], _np_ptr(K), _np_ptr(w2c), h, w, Hup, Wup,
                                          _p(zbuf), _stream())
    _lib.check(rc, "surf_raster_first_hit")
    return torch.where(zbuf == -1, torch.full_like(zbuf, -1), zbuf & 0xffffffff)


def marching_cubes(u, isovalue=0.0):
    """mcubes.marching_cubes(u, isovalue) (implicit_surface.py:353) on a device lattice u (nx, ny, nz) fp32.
    Returns (vertices (nv, 3) float64, triangles (nt, 3) int32) device tensors, vertices in lattice-index units.
    Two host syncs size the outputs (number of active lattice points, then vertex / triangle totals)."""
    _chk(u, torch.float32, "u")
    assert u.dim() == 3
    nx, ny, nz = (int(v) for v in u.shape)
    dev = u.device
    L = _lib.lib()
    flags = torch.empty(nx * ny * nz, dtype=torch.uint8, device=dev)
    iso = ctypes.c_double(float(isovalue))       # PyMCubes takes the isovalue as a double
    _lib.check(L.surf_mc_classify(_p(u), nx, ny, nz, iso, _p(flags), _stream()), "surf_mc_classify")
    active = compact(flags)
    m = int(active.shape[0])
    if m == 0:
        return torch.zeros(0, 3, dtype=torch.float64, device=dev), torch.zeros(0, 3, dtype=torch.int32, device=dev)
    ws = torch.empty(L.surf_mc_workspace_ints(m), dtype=torch.int32, device=dev)
    totals = torch.empty(2, dtype=torch.int32, device=dev)
    _lib.check(L.surf_mc_count(_p(flags), _p(active), m, _p(ws), _p(totals), _stream()), "surf_mc_count")
    n_v, n_t = (int(v) for v in totals.tolist())
    vertices = torch.empty(n_v, 3, dtype=torch.float64, device=dev)
    triangles = torch.empty(n_t, 3, dtype=torch.int32, device=dev)
    vbase = torch.empty(nx * ny * nz, dtype=torch.int32, device=dev)
    _lib.check(L.surf_mc_emit(_p(u), nx, ny, nz, iso, _p(flags), _p(active), m, _p(ws), _p(vbase), _p(vertices), _p(triangles),
                              _stream()), "surf_mc_emit")
    return vertices, triangles


# ------------------------------------------------------------------------------------------------
# sparse 3D U-Net pieces (reg_network.py)
# ------------------------------------------------------------------------------------------------

SUBM, DOWN, UP = 0, 1, 2


def table_from_coords(coords, D):
    table = torch.full((D, D, D), -1, dtype=torch.int32, device=coords.device)
    _lib.check(_lib.lib().surf_table_from_coords(_p(coords), coords.shape[0], int(D), _p(table), _stream()),
               "surf_table_from_coords")
    return table


# SURF_DOWN_DILATE / SURF_DOWN_FLOOR; "pad0" is the dilate kernel on coordinates stored + 1 with a fixed site range (below)
DOWN_RULES = {"dilate": 0, "floor": 1, "pad0": 0}
# The rule a conf without `reg_network.down_rule` gets (round 5): the reference pins torchsparse 2.1.0 (requirements.txt:205) and
# passes no padding (reg_network.py:9-13); that version's strided map is the spconv-style one = "pad0".
DEFAULT_DOWN_RULE = "pad0"
_pad0_ranges = {}


def down_sites(coords, D, rule=DEFAULT_DOWN_RULE, q_max=None):
    """Output sites of a k3/s2 sparse conv on the (D//2+1)^3 lattice: (coords2 (M,3) int32, table2, D2).
    rule: "dilate" (torchsparse's spdownsample for kernel != stride) or "floor" (unique(floor(c/2))): SURVEY App. C.
    "pad0" (the default): the spconv-style map with no padding (window 2q + {0,1,2}^3).  With every level's coordinates
    STORED + 1 (SparseCostRegNet.forward) that window is the centred one of the stored coordinates - 2 (q + 1) + {-1,0,1} =
    (2q + {0,1,2}) + 1 - so the same kernels serve; what differs is which even sites are outputs: all stored q in
    [1, q_max] with an input in the window (q_max = the true output lattice size (D_true - 3) // 2 + 1), instead of the even
    sites inside the inputs' bounding box."""
    _chk(coords, torch.int32, "coords")
    dev = coords.device
    D2 = D // 2 + 1
    if coords.shape[0] == 0:
        return (torch.empty(0, 3, dtype=torch.int32, device=dev), torch.full((D2, D2, D2), -1, dtype=torch.int32, device=dev), D2)
    if rule == "pad0":
        assert q_max is not None and q_max + 1 <= D2, (q_max, D2)
        key = (int(q_max), dev)
        if key not in _pad0_ranges:                                 # even sites 2 .. 2 q_max of the stored fine lattice
            _pad0_ranges[key] = torch.tensor([2, 2, 2, 2 * q_max, 2 * q_max, 2 * q_max], dtype=torch.int32, device=dev)
        bbox = _pad0_ranges[key]
    else:
        bbox = torch.empty(6, dtype=torch.int32, device=dev)          # stays on the device: no host round trip
        _lib.check(_lib.lib().surf_coords_bbox(_p(coords), coords.shape[0], _p(bbox), _stream()), "surf_coords_bbox")
    marks = torch.zeros(D2 * D2 * D2, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().surf_mark_down_sites(_p(coords), coords.shape[0], int(D), _p(bbox), _p(marks), DOWN_RULES[rule],
                                               _stream()), "surf_mark_down_sites")
    keys = compact(marks)
    c2 = torch.empty(keys.shape[0], 3, dtype=torch.int32, device=dev)
    t2 = torch.full((D2, D2, D2), -1, dtype=torch.int32, device=dev)
    if keys.shape[0] == 0:          # degenerate tiny lattices: no even site inside the inputs' bounding box
        return c2, t2, D2
    _lib.check(_lib.lib().surf_sites_from_keys(_p(keys), keys.shape[0], D2, _p(c2), _p(t2), _stream()), "surf_sites_from_keys")
    return c2, t2, D2


def spconv_pack_weights(weight, thin=False):
    """(27, Cin, Cout) fp32 device kernel -> split bf16 operand image of surf_spconv_mfma, or None when the channel pair
    has no matrix-core kernel.  Pairs with Cin or Cout < 16 (the finest lattices) are only packed with thin=True: their
    matrix-core form pays under the bf16 training policy alone (one product per offset; csrc/spconv_mfma.hip)."""
    _chk(weight, torch.float32, "weight")
    cin, cout = int(weight.shape[1]), int(weight.shape[2])
    if min(cin, cout) < 16 and not thin:
        return None
    nbytes = _lib.lib().surf_spconv_packed_bytes(cin, cout)
    if nbytes == 0:
        return None
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    _lib.check(_lib.lib().surf_spconv_pack_weights(_p(weight), cin, cout, _p(packed), _stream()), "surf_spconv_pack_weights")
    return packed


def spconv(x, in_table, out_coords, mode, weight, bn_scale=None, bn_shift=None, skip=None, packed=None, bf16=False):
    """One sparse conv + BN(eval) + ReLU (+ skip).  x (n_in, Cin); weight (27, Cin, Cout); returns (n_out, Cout).
    packed (spconv_pack_weights(weight)): run the matrix-core kernel instead of the per-voxel one; bf16 (with packed): operands
    rounded to bf16, one product per k-step (the training policy train_precision = bf16; never used by inference)."""
    _chk(x, torch.float32, "x")
    _chk(in_table, torch.int32, "in_table")
    _chk(out_coords, torch.int32, "out_coords")
    _chk(weight, torch.float32, "weight")
    cin, cout = int(weight.shape[1]), int(weight.shape[2])
    assert x.shape[1] == cin and weight.shape[0] == 27
    out = torch.empty(out_coords.shape[0], cout, dtype=torch.float32, device=x.device)
    if out_coords.shape[0] == 0:
        return out
    if x.shape[0] == 0:             # no input rows: every lookup misses, the output is relu(bn_shift) (+ skip)
        x = torch.zeros(1, cin, dtype=torch.float32, device=x.device)
    if packed is not None:
        _chk(packed, torch.uint8, "packed weights")
        assert packed.numel() == _lib.lib().surf_spconv_packed_bytes(cin, cout)
        rc = _lib.lib().surf_spconv_mfma(_p(x), cin, _p(in_table), int(in_table.shape[0]), _p(out_coords), out_coords.shape[0],
                                         int(mode), _p(packed), cout, _p(bn_scale), _p(bn_shift), _p(skip), _p(out), int(bool(bf16)),
                                         _stream())
        _lib.check(rc, "surf_spconv_mfma")
        return out
    if bf16 and bf16_rows and cin == 16 and cout == 8 and bn_scale is None and skip is None and (mode == SUBM or bf16_rows_all_modes):
        # the bf16 training policy: gather from the rows' bf16 shadow (attached by bn_train_relu / bn_relu_backward, or made here)
        r16 = getattr(x, "_rows16", None)
        if r16 is None:
            r16 = rows_to_bf16(x)
        rc = _lib.lib().surf_spconv_rows16(_p(r16), cin, _p(in_table), int(in_table.shape[0]), _p(out_coords), out_coords.shape[0],
                                           int(mode), _p(weight), cout, _p(out), _stream())
        _lib.check(rc, "surf_spconv_rows16")
        return out
    rc = _lib.lib().surf_spconv(_p(x), cin, _p(in_table), int(in_table.shape[0]), _p(out_coords), out_coords.shape[0], int(mode),
                                _p(weight), cout, _p(bn_scale), _p(bn_shift), _p(skip), _p(out), _stream())
    _lib.check(rc, "surf_spconv")
    return out


count_pairs = False     # bench.py (set_count_pairs): count the (site, offset) pairs of every sparse-conv backward during its warm-up steps
_pair_counts = {}


def set_count_pairs(on):
    """Switch the pair counting of spconv_backward on / off.  The counts are cached per (mode, lattice, rows in, rows out) for
    ONE scene: switching it on forgets what an earlier scene left (two scenes whose sizes coincide must not share a count)."""
    global count_pairs
    if on:
        _pair_counts.clear()
    count_pairs = bool(on)


def dgrad_weights(weight, mode, use_mfma=True, thin=False):
    """Kernel of the input-gradient convolution of a sparse convolution with `weight` (27, Cin, Cout): slices transposed
    (27, Cout, Cin), offsets mirrored for the submanifold mode; + its split-bf16 operand image for surf_spconv_mfma (None when
    use_mfma is off or a channel count is below 16)."""
    wt = weight.transpose(1, 2).contiguous()
    if mode == SUBM:
        wt = wt.flip(0).contiguous()
    return wt, (spconv_pack_weights(wt, thin) if use_mfma else None)


wgrad_up_from_coarse = True     # False: the transposed layers' weight gradient walks the fine sites (the form until round 5; A/B)


def spconv_backward(x, in_table, in_coords, out_table, out_coords, mode, weight, dy, use_mfma=True, dgrad=None, on_side=False):
    """Backward of y = spconv(x, in_table, out_coords, mode, weight) (no BN / ReLU / skip).  Returns (dx (n_in, Cin),
    dW (27, Cin, Cout)).  dx is a sparse convolution of dy over the OUTPUT lattice (`out_table` indexes out_coords):
    submanifold with mirrored offsets, down <-> up, kernel slices transposed.
    use_mfma: the wide layers' input gradient on the matrix cores like their forward (surf_spconv_mfma, bf16x3: both channel
    counts >= 16; the per-voxel kernel sat at 0.02-0.18 of HBM there, round 3); False: every layer on the fp32 per-voxel kernel
    (SparseCostRegNet.use_mfma).  dgrad: a cached dgrad_weights(weight, mode, use_mfma) (SparseCostRegNet keeps one per block
    and parameter version).  on_side: the weight gradient is launched on `side` (SideStream) - the caller calls side.join()
    before it reads dW."""
    _chk(dy, torch.float32, "dy")
    cin, cout = int(weight.shape[1]), int(weight.shape[2])
    wt, wt_packed = dgrad if dgrad is not None else dgrad_weights(weight, mode, use_mfma)
    pairs = 0
    key = (int(mode), int(in_table.shape[0]), int(x.shape[0]), int(out_coords.shape[0]))
    if count_pairs and key not in _pair_counts and out_coords.shape[0] > 0 and x.shape[0] > 0:
        # bench only, OUTSIDE its timed region (warm-up steps; the lattices of the bench scene are the same every step): the
        # number of (output site, offset) pairs that exist = the forward convolution of an all-ones 8-channel input with a
        # constant kernel (1/8), summed; a device scalar
        ones = torch.ones(x.shape[0], 8, dtype=torch.float32, device=x.device)
        _pair_counts[key] = spconv(ones, in_table, out_coords, mode,
                                   torch.full((27, 8, 8), 0.125, dtype=torch.float32, device=x.device))[:, 0].sum()
    if kernel_events is not None:
        pairs = _pair_counts.get(key, 0)

    def dgrad_():
        with _timed(f"spconv_dgrad<{cout},{cin}>", {"pairs": pairs, "sites": int(in_coords.shape[0])}):
            return spconv(dy, out_table, in_coords, {SUBM: SUBM, DOWN: UP, UP: DOWN}[mode], wt, packed=wt_packed,
                          bf16=colgram_precision == 1)

    if out_coords.shape[0] == 0 or x.shape[0] == 0:
        return dgrad_(), small_zeros(weight.shape, weight.device)
    # the weight gradient of a TRANSPOSED layer is computed from the coarse side (round 5): fine site c takes from coarse site q
    # through offset o when c = 2 q + o, which is the stride-2 DOWN relation with the lattices' roles swapped - so
    # dW[k][ci][co] = sum_q x[q][ci] dy[fine(2 q + o_k)][co] is the DOWN-mode weight gradient of (input = dy on the fine lattice,
    # output sites = the coarse ones, upstream = x), transposed.  Walking the FINE sites instead, seven of eight (site, offset)
    # references are ruled out by parity and only found absent after the lookup: 8 x the sites for the same sum.
    swap = mode == UP and wgrad_up_from_coarse
    if swap:
        wx, wtab, wcoords, wmode, wdy, wci, wco = dy, out_table, in_coords, DOWN, x, cout, cin
    else:
        wx, wtab, wcoords, wmode, wdy, wci, wco = x, in_table, out_coords, mode, dy, cin, cout
    dW = small_zeros((27, wci, wco), weight.device)       # the kernels accumulate into it

    def wgrad():
        with _timed(f"spconv_wgrad<{cin},{cout}>", {"pairs": pairs, "sites": int(out_coords.shape[0])}):
            if use_mfma and _lib.lib().surf_spconv_wgrad_mfma_supported(wci, wco):
                # round 5: the channel pairs for which the matrix-core form wins (spconv_wgrad_mfma.hip), fp32-equivalent
                rc = _lib.lib().surf_spconv_wgrad_mfma(_p(wx), wci, _p(wtab), int(wtab.shape[0]), _p(wcoords), wcoords.shape[0],
                                                       int(wmode), _p(wdy), wco, _p(dW), _stream())
            else:
                rc = _lib.lib().surf_spconv_wgrad(_p(wx), wci, _p(wtab), int(wtab.shape[0]), _p(wcoords), wcoords.shape[0],
                                                  int(wmode), _p(wdy), wco, _p(dW), _stream())
        _lib.check(rc, "surf_spconv_wgrad")

    if on_side:     # a leaf of the sweep: on the side stream, forked BEFORE the input gradient is issued so that the two run
        with side.fork():       # side by side (the caller joins before it reads dW)
            wgrad()
        side.keep(wx, wtab, wcoords, wdy, dW, lane=0)
        dx = dgrad_()
    else:
        dx = dgrad_()
        wgrad()
    if swap:
        dW = dW.transpose(1, 2)
    return dx, dW


def bn_train_relu(x, bn, skip=None, saved=None, counters=None, shadow=False):
    """spnn.BatchNorm in train mode + ReLU (+ skip) on raw convolution outputs x (n, C): batch statistics, running
    statistics updated in place like torch (reg_network.py:14-15,28-29).  bn: the block's nn.BatchNorm1d.
    saved: a dict that receives what bn_relu_backward needs (scale, shift, stats = mean | invstd).
    counters: a list that receives bn.num_batches_tracked INSTEAD of its increment - the caller bumps all of a network's
    counters with one torch._foreach_add_ (ten launches less per U-Net)."""
    _chk(x, torch.float32, "x")
    n, C = x.shape
    dev = x.device
    out = torch.empty_like(x)
    if n == 0:
        return out
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    shift = torch.empty(C, dtype=torch.float32, device=dev)
    stats = torch.empty(2 * C, dtype=torch.float32, device=dev) if saved is not None else None
    ws = torch.empty(_lib.lib().surf_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)
    momentum = 0.1 if bn.momentum is None else float(bn.momentum)
    track = bn.track_running_stats and bn.running_mean is not None
    rc = _lib.lib().surf_bn_train_affine(_p(x), n, C, _p(bn.weight.detach()), _p(bn.bias.detach()), float(bn.eps), momentum,
                                         _p(bn.running_mean if track else None), _p(bn.running_var if track else None),
                                         _p(scale), _p(shift), _p(stats), _p(ws), _stream())
    _lib.check(rc, "surf_bn_train_affine")
    if track:
        if counters is not None:
            counters.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
    if saved is not None:
        saved.update(scale=scale, shift=shift, stats=stats)
    out16 = torch.empty(n, C, dtype=torch.int16, device=dev) if shadow else None     # shadow: the rows' bf16 copy (ops.bf16_rows)
    _lib.check(_lib.lib().surf_bn_relu_apply16(_p(x), n, C, _p(scale), _p(shift), _p(skip), _p(out), _p(out16), _stream()),
               "surf_bn_relu_apply16")
    if shadow:
        out._rows16 = out16
    return out


def bn_relu_backward(x, dy, scale, shift, stats, train=True, shadow=False):
    """Backward of bn_train_relu (train) / of the folded eval-mode BN + ReLU (train=False): x the raw convolution output
    (n, C), dy the gradient of the block output (which is also the skip's gradient), scale / shift the forward's affine,
    stats = mean | invstd (2C).  Returns (dx (n, C), dgamma (C), dbeta (C))."""
    _chk(x, torch.float32, "x")
    _chk(dy, torch.float32, "dy")
    n, C = x.shape
    dev = x.device
    dx = torch.empty_like(x)
    if n == 0:
        return dx, small_zeros((C,), dev), small_zeros((C,), dev)
    dgb = torch.empty(2, C, dtype=torch.float32, device=dev)       # bn_bwd_finalize_kernel writes (not accumulates) both rows
    dgamma, dbeta = dgb[0], dgb[1]
    ws = torch.empty(_lib.lib().surf_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)
    mean, invstd = stats[:C], stats[C:]
    dx16 = torch.empty(n, C, dtype=torch.int16, device=dev) if shadow else None
    rc = _lib.lib().surf_bn_relu_backward16(_p(x), _p(dy), n, C, _p(scale), _p(shift), _p(mean), _p(invstd), int(bool(train)),
                                            _p(ws), _p(dgamma), _p(dbeta), _p(dx), _p(dx16), _stream())
    _lib.check(rc, "surf_bn_relu_backward16")
    if shadow:
        dx._rows16 = dx16
    return dx, dgamma, dbeta


def row_linear8(x, weight):
    _chk(x, torch.float32, "x")
    _chk(weight, torch.float32, "weight")
    assert x.shape[1] == 8 and tuple(weight.shape) == (8, 8)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().surf_row_linear8(_p(x), _p(weight), x.shape[0], _p(out), _stream()), "surf_row_linear8")
    return out


# ------------------------------------------------------------------------------------------------
# FPN pieces (feature_network.py)
# ------------------------------------------------------------------------------------------------


def conv3x3(x, w_packed, cout, stride=1, precision=0):
    """x (N,H,W,Cin) NHWC, w_packed [3][3][Cin][Cout] -> (N,H/stride,W/stride,Cout).  precision (layers with Cin >= 16, which run
    on the matrix cores): 0 = fp32-equivalent (exact bf16x3 split), 1 = operands rounded to bf16 (the bf16 training policy)."""
    _chk(x, torch.float32, "x")
    _chk(w_packed, torch.float32, "weight")
    N, H, W, cin = x.shape
    out = torch.empty(N, H // stride, W // stride, cout, dtype=torch.float32, device=x.device)
    rc = _lib.lib().surf_conv3x3_p(_p(x), _p(w_packed), N, H, W, cin, cout, stride, _p(out), int(precision), _stream())
    _lib.check(rc, f"surf_conv3x3({cin}->{cout}, stride {stride})")
    return out


def deconv3x3_s2(x, w_packed, cout, precision=0):
    _chk(x, torch.float32, "x")
    _chk(w_packed, torch.float32, "weight")
    N, H, W, cin = x.shape
    out = torch.empty(N, 2 * H, 2 * W, cout, dtype=torch.float32, device=x.device)
    rc = _lib.lib().surf_deconv3x3_s2_p(_p(x), _p(w_packed), N, H, W, cin, cout, _p(out), int(precision), _stream())
    _lib.check(rc, f"surf_deconv3x3_s2({cin}->{cout})")
    return out


def conv3x3_wgrad(big, small, stride=1, precision=0):
    """out[ky][kx][cb][cs] = sum big[n][ys S + ky - 1][xs S + kx - 1][cb] small[n][ys][xs][cs]: the weight gradient of a 3x3
    convolution (big = input, small = d output) or of the stride-2 transposed one (big = d output, small = input)."""
    _chk(big, torch.float32, "big")
    _chk(small, torch.float32, "small")
    N, Hs, Ws, cs = small.shape
    cb = big.shape[3]
    assert tuple(big.shape[:3]) == (N, Hs * stride, Ws * stride)
    ws = torch.empty(_lib.lib().surf_conv3x3_wgrad_workspace_floats(N, Hs, Ws, cb, cs), dtype=torch.float32, device=big.device)
    out = torch.empty(3, 3, cb, cs, dtype=torch.float32, device=big.device)
    rc = _lib.lib().surf_conv3x3_wgrad_p(_p(big), _p(small), N, Hs, Ws, cb, cs, int(stride), _p(ws), _p(out), int(precision), _stream())
    _lib.check(rc, f"surf_conv3x3_wgrad({cb}x{cs}, stride {stride})")
    return out


def inorm_relu_backward(raw, dy, stats):
    """Backward of inorm_relu_ (the skip's gradient is dy itself): raw (N,H,W,C) the convolution output before the in-place
    normalisation, stats (N,C,2) = mean | rstd.  One call for all views (surf_inorm_relu_backward: blockIdx.y = view; until
    round 5 a Python loop of surf_bn_relu_backward per view with sliced statistics: ~600 small launches per training step)."""
    _chk(raw, torch.float32, "raw")
    _chk(dy, torch.float32, "dy")
    _chk(stats, torch.float32, "stats")
    N, H, W, C = raw.shape
    assert tuple(dy.shape) == tuple(raw.shape) and tuple(stats.shape) == (N, C, 2)
    if C not in (8, 16, 32, 64):
        raise ValueError(f"inorm_relu_backward: C = {C} channels; instantiated for C in (8, 16, 32, 64)")
    dx = torch.empty_like(raw)
    ws = torch.empty(_lib.lib().surf_inorm_backward_workspace_bytes(N, C), dtype=torch.uint8, device=raw.device)
    rc = _lib.lib().surf_inorm_relu_backward(_p(raw), _p(dy), N, H * W, C, _p(stats), _p(ws), _p(dx), _stream())
    _lib.check(rc, "surf_inorm_relu_backward")
    return dx


def inorm_relu_(x, skip=None, want_stats=False, in_place=True):
    """x = relu(instance_norm(x)) (+ skip); x (N,H,W,C) NHWC.  in_place (default): x is overwritten and returned; False: x is
    left as it is (a recording forward keeps it as the raw convolution output) and a new tensor is returned."""
    _chk(x, torch.float32, "x")
    N, H, W, C = x.shape
    if C not in (4, 8, 16, 32, 64):
        raise ValueError(f"inorm_relu_: C = {C} channels; the FPN kernels are instantiated for C in (4, 8, 16, 32, 64) "
                         "(feature_network d_base = 8; include/surf_hip.h: surf_inorm_relu)")
    ws = torch.empty(_lib.lib().surf_inorm_workspace_doubles(N, H, W, C), dtype=torch.float64, device=x.device)
    stats = torch.empty(N, C, 2, dtype=torch.float32, device=x.device)
    out = x if in_place else torch.empty_like(x)
    _lib.check(_lib.lib().surf_inorm_relu_out(_p(x), N, H, W, C, _p(skip), _p(ws), _p(stats), _p(out), _stream()),
               "surf_inorm_relu_out")
    return (out, stats) if want_stats else out


# ------------------------------------------------------------------------------------------------
# small fused helpers of the training step (csrc/train_small.hip)
# ------------------------------------------------------------------------------------------------

def occupied_any(pts, volumes):
    """lookup_volume(pts, mask_volumes, 'nearest').any(-1) (implicit_surface.py:175) in one launch: (n,) bool."""
    _chk(pts, torch.float32, "pts")
    assert pts.dim() == 2 and pts.shape[1] == 3
    out = torch.empty(pts.shape[0], dtype=torch.bool, device=pts.device)
    if pts.shape[0] == 0:
        return out
    rc = _lib.lib().surf_occupied_any(_p(pts), pts.shape[0], volumes._tp, volumes._dp, volumes.n, _p(out), _stream())
    _lib.check(rc, "surf_occupied_any")
    return out


_l1_counters = {}


def _l1_mask(mask, ref):
    """(tensor or None, mask_kind of surf_masked_l1) for a mask given as a float / bool tensor or the string "target>0"."""
    if isinstance(mask, str):
        assert mask == "target>0"
        return None, 2
    assert mask.numel() == ref.numel() and mask.device == ref.device
    if mask.dtype == torch.bool or mask.dtype == torch.uint8:
        return mask.contiguous(), 1
    return _chk(mask.float().contiguous(), torch.float32, "mask"), 0


def masked_l1(pred, target, mask):
    """sum(|pred - target| mask) / (sum(mask) + 1e-8) as ONE launch (loss.py:71-93; fp64 sums in a fixed order).  mask: a float /
    bool tensor of pred's size, or "target>0".  Returns out2 (2,) fp32: [0] the loss, [1] 1 / (sum(mask) + 1e-8)."""
    _chk(pred, torch.float32, "pred")
    _chk(target, torch.float32, "target")
    assert pred.numel() == target.numel() and pred.numel() > 0
    dev = pred.device
    m, kind = _l1_mask(mask, pred)
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    if key not in _l1_counters:        # one self-resetting arrival counter per stream
        _l1_counters[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.lib().surf_masked_l1_workspace_bytes(), dtype=torch.uint8, device=dev)
    out2 = torch.empty(2, dtype=torch.float32, device=dev)
    rc = _lib.lib().surf_masked_l1(_p(pred), _p(target), _p(m), kind, pred.numel(), _p(ws), _p(_l1_counters[key]), _p(out2), _stream())
    _lib.check(rc, "surf_masked_l1")
    return out2


def masked_l1_backward(pred, target, mask, out2, upstream):
    """d (upstream * masked_l1) / d pred; upstream: a 0-d fp32 device tensor."""
    m, kind = _l1_mask(mask, pred)
    up = _chk(upstream.reshape(1).float().contiguous(), torch.float32, "upstream")
    g = torch.empty_like(pred)
    rc = _lib.lib().surf_masked_l1_backward(_p(pred), _p(target), _p(m), kind, pred.numel(), _p(out2), _p(up), _p(g), _stream())
    _lib.check(rc, "surf_masked_l1_backward")
    return g


def weight_norm_backward(vs, gs, dWs):
    """Backward of W_l = g_l v_l / |v_l|_row (sdf_network.py:88-89) for a list of layers in ONE launch: ([dv_l], [dg_l]),
    dg_l shaped like g_l."""
    n = len(vs)
    assert n == len(gs) == len(dWs) and n >= 1
    for v, g, dW in zip(vs, gs, dWs):
        _chk(v, torch.float32, "weight_v")
        _chk(g, torch.float32, "weight_g")
        _chk(dW, torch.float32, "dW")
        assert v.dim() == 2 and dW.shape == v.shape and g.numel() == v.shape[0]
    dvs = [torch.empty_like(v) for v in vs]
    dgs = [torch.empty_like(g) for g in gs]
    rows = (ctypes.c_int * n)(*[int(v.shape[0]) for v in vs])
    cols = (ctypes.c_int * n)(*[int(v.shape[1]) for v in vs])
    rc = _lib.lib().surf_weight_norm_backward(n, _ptr_array(vs), _ptr_array(gs), _ptr_array(dWs), rows, cols, _ptr_array(dvs),
                                              _ptr_array(dgs), _stream())
    _lib.check(rc, "surf_weight_norm_backward")
    return dvs, dgs
