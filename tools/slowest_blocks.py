import sys, os, ctypes as C, numpy as np, subprocess
sys.path.insert(0,'.')
import torch
from pgdrive_amd import _abi, bank, mapdata, scenario, build
lib = os.path.join("gpurun_out", "libpgd_prof.so")
subprocess.check_call([build.hipcc(), '--offload-arch=gfx950',*build.OPT,'-std=c++17',*build.FAST_FP,'-shared','-fPIC','-DPGD_PROF','-o',lib, build.SRC])
from pgdrive_amd import engine
L = engine.load_library(path=lib); engine._LIBH = L
descs = bank.load_descriptions()
mb = mapdata.MapBank(descs); sb = scenario.ScenarioBank(descs,[d['seed'] for d in descs])
N=int(sys.argv[1]) if len(sys.argv)>1 else 4096
cfg=_abi.make_config(N)
eng = engine.Engine(cfg, mb, sb)
eng.reset(np.arange(N)%100)
rng=np.random.default_rng(0)
acts = torch.from_numpy(rng.uniform(-1,1,size=(64,N,1,2)).astype(np.float32)).cuda()
names=['load','trig+snap','policy','dynamics','crash','after_step','reward','reset','store','i_route','i_search','i_lc','i_pid','ld_stage','obs','WALL','as_route','as_getlane','as_local','as_side','o_pub','o_compact','o_state','o_neigh','o_lidar']
raw=(C.c_ulonglong*(N*32))()
L.pgd_debug_phase_raw.argtypes=[C.c_void_p, C.c_void_p, C.c_int]
with torch.cuda.stream(eng.stream):
    for k in range(1500): eng.step(acts[k%64])
    L.pgd_debug_phase_raw(eng.h, raw, N)
    for k in range(6):
        f,i,ei=eng.get_state()
        nact=(i[0,:,1:]==2).sum(1); npend=(i[0,:,1:]==1).sum(1)
        o,r,d,fl=eng.step(acts[k%64]); eng.sync()
        L.pgd_debug_phase_raw(eng.h, raw, N)
        a=np.frombuffer(raw,dtype=np.uint64).reshape(N,32).astype(np.int64)
        wall=a[:,15]
        order=np.argsort(-wall)
        dn=d.cpu().numpy().reshape(-1)
        print('step',k,'wall us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f'%(wall.mean()/100,np.percentile(wall,50)/100,np.percentile(wall,90)/100,np.percentile(wall,99)/100,wall.max()/100))
        for b in order[:6]:
            ph={n:int(a[b,j]) for j,n in enumerate(names) if a[b,j]>1500 and n!='WALL'}
            print('   blk',b,'wall',wall[b]/100,'done',int(dn[b]),'nact',int(nact[b]),'npend',int(npend[b]),ph)
        # mean wall by class
        for lab,m in (('done',dn==1),('nact0',(nact==0)&(dn==0)),('nact1-2',(nact>0)&(nact<3)&(dn==0)),('nact3+',(nact>=3)&(dn==0))):
            if m.sum(): print('   class',lab,'n',int(m.sum()),'mean wall %.1f max %.1f'%(wall[m].mean()/100,wall[m].max()/100))
