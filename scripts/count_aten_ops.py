"""Which Python lines of the training step launch torch's own helper kernels (fill / copy / add / reduce ...)?

Runs `bench.training_step_setup` + two warm steps, then ONE step under a TorchDispatchMode that books every aten op to the
innermost frame inside surf_amd/ (or bench.py) on the Python stack - forward and autograd-backward threads alike.  Prints
`count  op  file:line` sorted by count, and the totals per op.  Diagnostic only (scripts/, not the product).

    python scripts/count_aten_ops.py [--small]        # --small: a reduced scene (runs on the CPU build's tests? no: needs the GPU)
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from surf_amd import training  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.permute",
        "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.select", "aten.slice", "aten.as_strided", "aten.unbind",
        "aten.split", "aten.empty", "aten.new_empty", "aten.narrow", "aten.unflatten", "aten.lift_fresh", "aten.is_", "aten.sym_",
        "aten.chunk", "aten.view_as", "aten.flatten", "aten.unfold", "aten.diagonal", "aten.movedim",
        "aten.empty_like", "aten.empty_strided", "aten.new_empty_strided", "aten.result_type", "aten.set_", "aten.stride",
        "aten.size", "aten._to_copy_view")


class Book(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=40)[:-1]):
                fn = fr.filename
                if ("surf_amd" in fn or fn.endswith("bench.py")) and "count_aten_ops" not in fn:
                    site = f"{os.path.relpath(fn, ROOT)}:{fr.lineno} {fr.name}"
                    break
            if name.startswith(("aten._to_copy", "aten.copy_", "aten._local_scalar_dense", "aten.clone", "aten.zeros", "aten.full",
                                "aten.add.", "aten.add_", "aten.cat")):
                # where the bytes move and how many: host<->device copies and large fills are what the launch count hides
                flat = []
                for a in list(args) + list((kwargs or {}).values()):
                    flat.extend(a if isinstance(a, (list, tuple)) else [a])
                ts = [a for a in flat if torch.is_tensor(a)]
                devs = "/".join(sorted({t.device.type for t in ts})) or "-"
                kw = kwargs or {}
                if "device" in kw and kw["device"] is not None:
                    devs += "->" + torch.device(kw["device"]).type
                numel = max([t.numel() for t in ts] + [0])
                if not ts and args and isinstance(args[0], (list, tuple)) and all(isinstance(d, int) for d in args[0]):
                    numel = 1
                    for d in args[0]:
                        numel *= d
                name += f" [{devs} {'big' if numel >= (1 << 20) else 'small'}]"
            self.sites[(name, site)] += 1
            self.ops[name] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device("cuda:0")
    small = "--small" in sys.argv
    kw = dict(H=144, W=200, base_dim=24, rays=128) if small else {}
    model, ipts, targets, loss_fn, opt = bench.training_step_setup(dev, **kw)
    for _ in range(2):
        training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
    torch.cuda.synchronize()
    book = Book()
    with book:
        training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
    torch.cuda.synchronize()
    print("== per op")
    for name, n in book.ops.most_common():
        print(f"{n:6d}  {name}")
    print("== per site")
    for (name, site), n in book.sites.most_common(400):
        print(f"{n:6d}  {name:58s} {site}")


if __name__ == "__main__":
    main()
