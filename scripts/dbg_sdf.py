import sys, torch, numpy as np
sys.path.insert(0, '.')
from tests.conftest import load_npz
from tests.golden_cfg import pipeline_views
from surf_amd import ops
from oracle import surf_oracle as O
d = torch.device('cuda:0')
w = load_npz('weights.npz'); gp = load_npz('pipeline.npz'); gr = load_npz('render.npz')
vols, tabs, masks, mvol = pipeline_views(gp)
sv = ops.SparseVolumes([v.to(d) for v in vols], [t.to(d) for t in tabs])
pk = ops.sdf_pack_weights(w, d)
pts = gr['pts']
sdf, grad = ops.sdf_mlp(pts.to(d).contiguous(), sv, pk)
g = grad.cpu(); ref = gr['sdf_grad']
err = (g - ref).abs().max(1).values
bad = (err > 1e-3).nonzero().view(-1)
print('bad idx', bad.tolist())
print('err of bad', err[bad][:20])
# which term: oracle pieces
layers = O.sdf_weights(w)
phi, jphi = O.lookup_sparse_volume(pts, vols, tabs, with_jac=True)
s0, g_full, _ = O.sdf_mlp(layers, pts, phi, jphi)
s1, g_nophi, _ = O.sdf_mlp(layers, pts, phi, torch.zeros_like(jphi))
print('jphi max per point (bad):', jphi.abs().amax((1,2))[bad][:10], 'good:', jphi.abs().amax((1,2))[:10])
print('diff vs oracle-without-Jphi for bad:', (g - g_nophi).abs().max(1).values[bad][:10])
inside = (pts.abs().max(1).values <= 1.0)
print('bad inside cube?', inside[bad].tolist())
# repeated single point
p = pts[200:201].repeat(4096, 1).contiguous()
s2, g2 = ops.sdf_mlp(p.to(d), sv, pk)
g2 = g2.cpu()
dev = (g2 - g2[0]).abs().max(1).values
print('repeat: n deviating', int((dev > 0).sum()), (dev > 0).nonzero().view(-1)[:40].tolist())
print('ref grad', ref[200], 'gpu', g2[0])
