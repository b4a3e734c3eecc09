import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from pgdrive_amd import _abi, bank, mapdata, scenario, engine
descs = bank.load_descriptions()
mb = mapdata.MapBank(descs); sb = scenario.ScenarioBank(descs,[d['seed'] for d in descs])
N=4096
cfg=_abi.make_config(N, seed=7)
eng = engine.Engine(cfg, mb, sb)
eng.reset((np.arange(N)%100).astype(np.int32))
g=torch.Generator(device='cuda'); g.manual_seed(0)
bad=0; ndone=0; t0=time.time()
mins=torch.full((1,),9.0,device='cuda'); maxs=torch.full((1,),-9.0,device='cuda'); nan=torch.zeros((1,),device='cuda')
with torch.cuda.stream(eng.stream):
    for k in range(30000):
        mode=(k//2000)%3
        a=torch.rand((N,1,2),device='cuda',generator=g)*2-1
        if mode==1: a[...,1]=1.0; a[...,0]*=0.05
        if mode==2: a[...,1]=a[...,1].abs()
        obs,rew,done,fl=eng.step(a)
        if k%50==0:
            nan+= (~torch.isfinite(obs)).sum()+(~torch.isfinite(rew)).sum()
            mins=torch.minimum(mins,obs.min().view(1)); maxs=torch.maximum(maxs,obs.max().view(1))
        ndone+=0
    eng.sync()
f,i,ei=eng.get_state()
print('steps',30000,'time',round(time.time()-t0,1),'nan',int(nan.item()),'obs range',float(mins.item()),float(maxs.item()))
print('episodes per env mean',ei[_abi.EI['EPISODES']].mean() if hasattr(_abi,'EI') else 'n/a','state finite',np.isfinite(f).all(), 'status hist',np.bincount(i[0].ravel(),minlength=5))
print('max |x|,|y|',np.abs(f[0]).max(),np.abs(f[1]).max(),'speed max',np.abs(f[3]).max())
