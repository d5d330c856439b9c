import sys, numpy as np, torch
sys.path.insert(0,'.')
import dispu_amd.tf_approxmatch as A
from dispu_amd import synth
from oracle import oracle as O
dev=torch.device('cuda:0')
for (b,n,m) in [(1,1100,1030),(2,300,100),(2,100,300),(1,1024,1024)]:
    x1,x2=synth.patches(b,n,seed=n),synth.patches(b,m,seed=m+1)
    t1,t2=torch.from_numpy(x1).to(dev),torch.from_numpy(x2).to(dev)
    mg=A.approx_match(t1,t2); mo=O.approx_match(x1,x2)
    d=np.abs(mg.cpu().numpy()-mo)
    cg=A.match_cost(t1,t2,mg).cpu().numpy(); co=O.match_cost(x1,x2,mo)
    print((b,n,m),'match max abs',d.max(),'n>1e-3',(d>1e-3).sum(),'n>1e-4',(d>1e-4).sum(),'cost rel',np.abs(cg-co)/co, 'mass', mg.sum().item(), mo.sum())
