"""Multi-GPU plumbing of the hot path.  Inference shards by scene: scenes are independent, every rank
renders its own share, there is NO data-path collective.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in
the CPU tests) is only used for the barrier / max-over-ranks timing of bench.py, to gather small per-scene result records
and for training (runner.py:102,163).  A train-mode forward is differentiable (surf_amd.autograd), so
`torch.nn.parallel.DistributedDataParallel(model)` works as in the reference: it broadcasts rank 0's parameters and buffers
when it wraps the model, rank 0's BatchNorm running statistics at every forward, and averages the gradients in its bucket
hooks during `loss.backward()`.  `all_reduce_gradients` + `broadcast_module_state` are the same two duties in explicit
form for a step that does not wrap the model (surf_amd.training, bench.py)."""
import os

import torch
import torch.distributed as dist


# True once init_from_env(force=True) has brought up a group at world size 1: the helpers below then ISSUE their collectives
# instead of short-circuiting, so that every RCCL call of an N-rank run (barrier, MAX all-reduce of a device tensor, the padded
# device all_gather of gather_rows, the flat gradient bucket, the parameter broadcast, DDP's bucket hooks) executes on a
# one-GPU box.  At world size 1 each of them is the identity on the data: results must be bit-equal to the no-group run
# (tests/test_rccl_world1.py).
FORCED = False


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """File descriptor 1 points at stderr inside the block (and C stdio is flushed on both edges).  librccl prints a version
    banner ("RCCL version : ... / Librccl path : ...") with printf to STDOUT when it creates its first communicator: on a
    stdout that carries a result line (bench.py prints exactly one JSON line) the banner must land on stderr instead."""
    import ctypes
    import sys
    try:
        libc = ctypes.CDLL(None)
    except OSError:
        libc = None
    sys.stdout.flush()
    if libc is not None:
        libc.fflush(None)
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        if libc is not None:
            libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def init_from_env(backend="nccl", device=None, force=False, timeout_s=None):
    """env:// rendezvous as launched by torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    A group is created when WORLD_SIZE > 1 (utils/distribute.py:66-88 of the reference) or, with `force`, at world size 1 too:
    one rank that is its own peer, rendezvous on a free 127.0.0.1 port.  For "nccl" (= RCCL) the caller must have called
    torch.cuda.set_device(device) before; the group is bound to it (device_id) so that the communicator is created eagerly."""
    global FORCED
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=timeout_s)
        with stdout_to_stderr():                        # RCCL's banner (communicator creation) -> stderr
            dist.init_process_group(backend, **kw)
            if backend == "nccl" and device is not None:
                dist.barrier(device_ids=[device.index if device.index is not None else torch.cuda.current_device()])
                torch.cuda.synchronize(device)
        FORCED = bool(force) and world == 1
    return rank, local_rank, world


def _active():
    """Collectives are issued when a group is up and it has peers - or when it was forced at world size 1."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCED)


def backend_note(one_gpu=False):
    """What bench.py prints as `collective_backend`."""
    if not dist.is_initialized():
        return None
    w = dist.get_world_size()
    lib = ""
    if dist.get_backend() == "nccl":
        try:
            lib = " = RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:       # noqa: BLE001 - version query only
            lib = " = RCCL"
    return f"{dist.get_backend()}{lib} (world {w}" + (", all ranks on cuda:0" if one_gpu else "") + ")"


def shard_scenes(n_scenes, rank, world):
    """Round-robin scene -> rank map (the reference's DistributedSampler without shuffling, datasets/__init__.py:37-38;
    no padding: a rank may get one scene fewer)."""
    return list(range(rank, n_scenes, world))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (the timing contract of bench.py)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(record, dst=0):
    """Collect one small picklable record per rank on `dst` (per-scene metrics); None elsewhere."""
    if not dist.is_initialized():
        return [record]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(record, out, dst=dst)
    return out


def gather_rows(t, dst=0):
    """Concatenate every rank's rows (dim 0, possibly different counts per rank) on `dst`, in rank order; None elsewhere.
    The data-path collective of the single-scene split (bench.py --split rays: rank r renders rays [r R / N, (r + 1) R / N) of ONE
    image, the image is stitched on rank 0).  RCCL: one padded all_gather of device tensors over xGMI; gloo (tests): through
    host copies (gloo has no device all_gather)."""
    if not _active():
        return t
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    counts = [None] * world
    dist.all_gather_object(counts, int(t.shape[0]))
    m = max(counts)
    if m == 0:                                 # nothing anywhere: no tensor collective (RCCL rejects empty buffers)
        return t if rank == dst else None
    src = t if backend == "nccl" else t.cpu()
    if src.shape[0] < m:
        pad = torch.zeros((m - src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        src = torch.cat([src, pad])
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src.contiguous())
    if rank != dst:
        return None
    return torch.cat([p[:c] for p, c in zip(parts, counts)]).to(t.device)


def all_reduce_gradients(params, bucket_bytes=25 << 20):
    """Average `.grad` over the ranks (what DistributedDataParallel does under loss.backward(), runner.py:102,163): the
    gradients are flattened into buckets of at most `bucket_bytes` (DDP's default 25 MB: the ~1.4 M parameters of SuRF, 5.6 MB,
    make ONE bucket - a single ring all-reduce over xGMI, latency-bound), summed with all_reduce and divided by the world
    size.  Parameters without a gradient contribute zeros so that every rank issues the same collectives."""
    if not _active():
        return 0
    params = [p for p in params if p.requires_grad]
    world = dist.get_world_size()
    buckets, cur, size = [], [], 0
    for p in params:
        nbytes = p.numel() * 4
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nbytes
    if cur:
        buckets.append(cur)
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        off = 0
        for p in bucket:
            g = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            # fp32 parameters take their slice of the reduced bucket AS their gradient (a view, like DDP's gradient_as_bucket_view:
            # no copy back - 150 small launches at the end of a step); other dtypes get a converted copy
            p.grad = g if p.dtype == torch.float32 else g.to(p.dtype)
    return len(buckets)



def broadcast_module_state(module, src=0, buffers_only=False):
    """Make every rank's parameters and buffers (BatchNorm running statistics) equal to rank `src`'s - what
    DistributedDataParallel does when it wraps a model (parameters + buffers once) and at every forward (buffers,
    `broadcast_buffers=True`, runner.py:102).  Without it replicas started from different seeds, or a checkpoint loaded on
    rank 0 only, stay different forever under a gradient-only all-reduce.  Call once before the first step and with
    `buffers_only=True` before evaluation / checkpointing (or every step, like DDP).  Returns the number of tensors sent."""
    if not _active():
        return 0
    tensors = list(module.buffers()) if buffers_only else list(module.parameters()) + list(module.buffers())
    n = 0
    with torch.no_grad():
        for t in tensors:
            if not torch.is_tensor(t) or t.numel() == 0:
                continue
            d = t.detach()                     # shares the version counter with t (t.data does not)
            dist.broadcast(d, src=src)
            # c10d writes through the storage without telling autograd: every weight cache of the package (packed SDF /
            # blend images, inv_s, agg_mlp host copy, sparse-conv prep, the frozen-volume scene) is keyed on
            # (parameter._version, data_ptr) and would keep serving the pre-broadcast weights.  A self copy_ is an in-place
            # op autograd sees: it bumps the shared counter (no value change, any dtype).
            d.copy_(d)
            n += 1
    return n


def shutdown():
    """Leave the process group in step: barrier, then destroy_process_group().  A rank that simply exits while another is
    still inside its last collective makes gloo's (and RCCL's) background thread abort the slower one
    ("terminate called without an active exception").  Every spawned worker and bench.py end with this."""
    global FORCED
    if dist.is_initialized():
        try:
            dist.barrier()
        finally:
            dist.destroy_process_group()
            FORCED = False
