import sys, torch, numpy as np, math
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
pts = gr['pts'][:32].contiguous()
sdf, grad = ops.sdf_mlp(pts.to(d), sv, pk)
torch.cuda.synchronize()
scr = list(ops._scratch_cache.values())[0].view(torch.float32).cpu().numpy()
SCR_S = 5*16*64*4
S = scr[:SCR_S].reshape(5, 16, 64, 4)      # [layer][t*4+g][lane][i]
Jg = scr[SCR_S:SCR_S+11*64*4].reshape(11, 64, 4)
# oracle pre-activations
layers = O.sdf_weights(w)
phi, jphi = O.lookup_sparse_volume(pts, vols, tabs, with_jac=True)
e = O.posenc(pts)
h = e; pre = []
for l,(W,b) in enumerate(layers):
    if l == 3: h = torch.cat([h, e], -1)/math.sqrt(2)
    if l > 0: h = torch.cat([h, phi], -1)
    t = h @ W.t() + b; pre.append(t)
    h = O.softplus100(t) if l < 6 else t
lanes = np.arange(64); J = lanes & 31; H = lanes >> 5
for l in range(5):
    sp = O.softplus100_grad(pre[l]).numpy()
    sp = np.concatenate([sp, np.full((32, 128 - sp.shape[1]), 0.5)], 1)
    exp = np.zeros((16, 64, 4))
    for t in range(4):
        for g in range(4):
            for i in range(4):
                r = 4*g+i
                k = 32*t + (r&3) + 8*(r>>2) + 4*H
                exp[t*4+g, :, i] = sp[J, k]
    err = np.abs(S[l] - exp).max(axis=(0, 2))
    print('layer', l, 'bad lanes', np.nonzero(err > 1e-4)[0].tolist(), 'max', err.max())
jexp = np.zeros((64, 44))
sel = 14*H[:, None] + np.arange(14)[None, :]
jj = jphi.numpy()[J]                      # (64,28,3)
jl = np.take_along_axis(jj, sel[:, :, None].repeat(3, 2), 1).reshape(64, 42)
jexp[:, :42] = jl
jgot = Jg.transpose(1, 0, 2).reshape(64, 44)
err = np.abs(jgot - jexp).max(1)
print('J bad lanes', np.nonzero(err > 1e-4)[0].tolist(), err.max())
g_ref = gr['sdf_grad'][:32]
print('grad err per point', (grad.cpu() - g_ref).abs().max(1).values)
print('--- relation')
l = 1
sp = O.softplus100_grad(pre[l]).numpy()
for lane in (12, 13, 44, 11):
    t_, g_ = 1, 2
    j, hh = lane & 31, lane >> 5
    for i in range(4):
        r = 4*g_+i
        k = 32*t_ + (r&3) + 8*(r>>2) + 4*hh
        tt = pre[l][j, k].item(); bt = 100*tt
        e_ = math.exp(min(bt, 20))
        print(lane, i, 'S=', S[l][t_*4+g_, lane, i], 'exp sp=', sp[j, k], 'bt=', bt, 'e=', e_, 'h=', math.log1p(e_)/100, 't=', tt)
