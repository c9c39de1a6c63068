import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from oracle import surf_oracle as O
from surf_amd import ops
from tests.golden_cfg import CFG, pipeline_views
g=lambda n: {k: torch.from_numpy(v) for k,v in np.load(f'tests/golden/{n}.npz').items()}
weights=g('weights'); gp=g('pipeline'); sc=g('scene'); gr=g('render'); fpn=g('fpn')
d=torch.device('cuda:0')
feats=[fpn[f"out{i}"] for i in range(4)][::-1]
feats_t4=[ops.pack_texel4(f.to(d).contiguous()) for f in feats]
imgs_t4=ops.pack_texel4(sc["imgs"].to(d).contiguous())
cams=ops.Cameras(sc["intrs"], sc["c2ws"])
pts=gr["pts"].clone(); n=pts.shape[0]
gen=torch.Generator().manual_seed(21)
gcolor=torch.randn(n,3,generator=gen)
idx=torch.arange(0,n,dtype=torch.int32)[torch.rand(n,generator=gen)>0.2].contiguous()
raw=torch.from_numpy(ops.blend_raw_weights(weights)).to(d)
res=ops.blend_backward(pts.to(d).contiguous(), idx.to(d), gcolor.to(d).contiguous(), feats_t4, imgs_t4, cams, raw, want_color=True)
prefix="implicit_surface.color_network."
sd={k: v.clone().requires_grad_(True) for k,v in weights.items() if k.startswith(prefix)}
rf,rdiff,mval=O.lookup_feature(pts[idx.long()], sc["imgs"], sc["intrs"], sc["c2ws"], feats)
col=O.blending(sd, rf, rdiff, mval)
print('color err', float((res["_color"].cpu()-col.detach()).abs().max()))
(col*gcolor[idx.long()]).sum().backward()
for k,v in sd.items():
    name=k[len(prefix):]; ref=v.grad if v.grad is not None else torch.zeros_like(v)
    got=res[name].reshape(ref.shape).cpu()
    print(f"{name:18s} ref max {float(ref.abs().max()):.3e} err {float((got-ref).abs().max()):.3e}")
# ties in min?
ex=torch.exp(sd[prefix+"s"].abs().detach()*(rdiff[...,3:4]-1))
print('valid views per point histogram', torch.bincount(mval.sum(1)))
# per-sample s gradients
import ctypes
V=cams.nv-1
pts_d=pts.to(d).contiguous()
# rerun to get ds per sample: call the ABI directly
from surf_amd import _lib
ROW=_lib.lib().surf_blend_backward_row_floats()
nA=int(idx.shape[0])
rows=torch.empty(nA,V,ROW,device=d); ds=torch.zeros(nA,device=d)
hw=(ctypes.c_int*8)(*[int(v) for f in feats_t4 for v in f.shape[1:3]])
intr16=np.ascontiguousarray(cams.intrs.reshape(cams.nv,-1))
rc=_lib.lib().surf_blend_backward(ops._p(pts_d), ops._p(idx.to(d)), nA, ops._p(gcolor.to(d).contiguous()), ops._ptr_array(list(feats_t4)), hw, ops._p(imgs_t4), cams.nv, ops._np_ptr(intr16), ops._np_ptr(cams.w2c), ops._np_ptr(cams.c2w), ops._p(raw), ops._p(rows), ops._p(ds), None, ops._stream())
torch.cuda.synchronize()
ds=ds.cpu()
sk=prefix+"s"
refs=[]
for j in range(nA):
    sd2={k: v.detach().clone().requires_grad_(k==sk) for k,v in weights.items() if k.startswith(prefix)}
    c=O.blending(sd2, rf[j:j+1], rdiff[j:j+1], mval[j:j+1])
    (c*gcolor[idx.long()][j:j+1]).sum().backward()
    refs.append(float(sd2[sk].grad))
refs=torch.tensor(refs)
err=(ds-refs).abs()
print('per-sample ds: max ref', float(refs.abs().max()), 'max err', float(err.max()), 'n bad', int((err>1e-6+1e-3*refs.abs()).sum()))
bad=torch.nonzero(err>1e-6+1e-3*refs.abs()).view(-1)[:10]
for j in bad.tolist():
    exv=torch.exp(weights[sk].abs()*(rdiff[j,:,3]-1))
    print(j, 'ds', float(ds[j]), 'ref', refs[j].item(), 'mask', mval[j].tolist(), 'ex', exv.tolist())
