import sys; sys.path.insert(0,'ransac-flow_amd'); sys.path.insert(0,'oracle')
import torch, numpy as np
import torch.nn.functional as TF
from rfx import ops, synth
dev='cuda'
n, hd, wd = 11, 60, 80
fd, fd2, pm, md = synth.assembly_arrays(31, n, hd, wd)
H,W=480,640
F=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
m11=ops.resize_bilinear(F(md),(H,W),False); m1=ops.resize_bilinear(F(md[:1]),(H,W),False)
ref=TF.interpolate(torch.from_numpy(md),size=(H,W),mode='bilinear',align_corners=False)
d=(m11[:1]-m1).abs()
print('diff count', int((d>0).sum()), 'of', d.numel())
nz=(d>0).nonzero()[:8]; print(nz.tolist())
print('m11 vs ref', float((m11.cpu()-ref).abs().max()), int(((m11.cpu()-ref).abs()>0).sum()))
print('m1 vs ref', float((m1.cpu()-ref[:1]).abs().max()), int(((m1.cpu()-ref[:1]).abs()>0).sum()))
m11b=ops.resize_bilinear(F(md),(H,W),False); print('repeat equal', torch.equal(m11,m11b))
