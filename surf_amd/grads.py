"""Where the HIP backward routines put parameter gradients.

The backward kernels produce plain tensors.  Called directly (`SuRF.backward`, `SuRF.backward_volumes`,
`ImplicitSurface.backward_render`: the explicit step of `surf_amd.training`) they accumulate into `.grad` like
`loss.backward()` does; called from a `torch.autograd.Function` (`surf_amd.autograd`: the reference's own
`loss.backward()`, runner.py:163) they must instead RETURN the gradients so that autograd's AccumulateGrad nodes - and
with them DistributedDataParallel's reduction hooks (runner.py:102) - see them.  A `GradSink` collects them per parameter.
"""
import torch


class GradSink:
    def __init__(self):
        self._g = {}

    def add(self, p, g):
        k = id(p)
        self._g[k] = g if k not in self._g else self._g[k] + g

    def get(self, p):
        """The collected gradient of `p`, zeros if the sweep never reached it (e.g. the blocks of an empty U-Net level)."""
        g = self._g.get(id(p))
        return torch.zeros_like(p) if g is None else g.reshape(p.shape)


def accumulate(p, g, sink=None):
    g = g.reshape(p.shape).to(device=p.device, dtype=p.dtype)
    if g.stride() != p.stride():        # e.g. one column of a wider buffer as a 1-element parameter's gradient: DistributedDataParallel
        g = g.clone(memory_format=torch.contiguous_format)      # wants the parameter's layout in its bucket views (it warns otherwise)
    if sink is not None:
        sink.add(p, g)
    else:
        p.grad = g if p.grad is None else p.grad + g
