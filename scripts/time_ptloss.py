"""Photometric terms (losses/photometric_loss.py:54-125) at the training shape: forward and backward launches timed with HIP events.
    python scripts/time_ptloss.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import ops
from bench import training_step_setup

dev = torch.device("cuda:0")
model, ipts, targets, loss_fn, opt = training_step_setup(dev, 576, 800, 5, 88, 512)
imgs_t4 = ops.pack_texel4(targets["imgs"].float().contiguous())
cams = ops.Cameras(targets["intrs"], targets["c2ws"])
H, W = imgs_t4.shape[1:3]
g = torch.Generator().manual_seed(1)
depth = (2.0 + 0.5 * torch.rand(H, W, generator=g)).to(dev)
mask = torch.ones(H, W, device=dev)
up = torch.tensor(0.5, device=dev)
for ref_idx, topk in ((0, 2), (1, 1)):
    for _ in range(3):
        loss, st = ops.photometric_loss(depth, imgs_t4, mask, cams, ref_idx=ref_idx, topk=topk, return_state=True)
        ops.photometric_loss_backward(depth, imgs_t4, mask, cams, ref_idx, topk, upstream=up, state=st)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    N = 20
    tf = tb = 0.0
    for _ in range(N):
        e[0].record()
        loss, st = ops.photometric_loss(depth, imgs_t4, mask, cams, ref_idx=ref_idx, topk=topk, return_state=True)
        e[1].record()
        ops.photometric_loss_backward(depth, imgs_t4, mask, cams, ref_idx, topk, upstream=up, state=st)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"ref_idx {ref_idx} topk {topk}: forward {tf / N * 1e3:.0f} us, backward {tb / N * 1e3:.0f} us (incl. their small torch reductions)")
