"""Time one full training step (runner.py:150-166, generalisation training: FPN -> 4-stage volume build -> render -> loss ->
HIP backward of everything -> Adam) of a volume-building SuRF on the bench scene (5 views, 576x800, 88^3 base lattice) for a
batch of 512 rays x 128 samples.  Weights are random-init: the analytic sphere logit replaces the U-Net's matching logit in
the forward (as in bench.py's volume-build timing) so that the voxel pyramid is the surface-concentrated one a trained
network produces; the backward still runs through every kernel."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import ops, training

from bench import training_step_setup

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nv, H, W = 5, int(sys.argv[2]) if len(sys.argv) > 2 else 576, int(sys.argv[3]) if len(sys.argv) > 3 else 800      # surf.conf trains at 480 640
model, ipts, targets, loss_fn, opt = training_step_setup(dev, H, W, nv, 88, R, device_jitter=os.environ.get("SURF_CPU_JITTER", "0") != "1")

# instrument the phases with HIP events by wrapping the model's entry points
marks = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        marks.append((label, e0, e1))
        return r
    setattr(obj, name, timed)


wrap(model.feature_network, "forward", "fwd  FPN")
wrap(model, "build_volumes", "fwd  volume build (4 stages)")
wrap(model.implicit_surface, "render_scene", "fwd  render (+ patch warp, H.1, sparse sdf)")
wrap(ops, "photometric_loss", "loss photometric terms (8 maps)")
wrap(ops, "photometric_loss_backward", "bwd  photometric terms")
wrap(model.implicit_surface, "backward_render", "bwd  render (composite, sdf, blend)")
wrap(model.feature_network, "backward", "bwd  FPN")
for s, net in enumerate(model.reg_network.nets):
    wrap(net, "backward", "bwd  sparse U-Net")
wrap(model.volume, "stage_backward", "bwd  cost volume + parent rows")
wrap(model.matching_field, "backward", "bwd  matching field")
wrap(ops, "densify_backward", "bwd  densify")
wrap(opt, "step", "Adam")

for _ in range(2):
    marks.clear()
    training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
N = 5
acc = {}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    marks.clear()
    out = training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
    torch.cuda.synchronize()
    for label, e0, e1 in marks:
        acc[label] = acc.get(label, 0.0) + e0.elapsed_time(e1)
wall = (time.perf_counter() - t0) / N * 1e3
nvox = model.last_voxels_per_stage
print(f"{R} rays x 128 samples, {nv} views {H}x{W}, voxels per stage {nvox}: wall {wall:.1f} ms per training step, loss {out['loss']:.4f}")
for k, v in acc.items():
    print(f"  {k}: {v / N:.2f} ms")
