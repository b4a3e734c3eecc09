import sys, os, ctypes as C, numpy as np, subprocess
sys.path.insert(0,'.')
import torch
from pgdrive_amd import _abi, bank, mapdata, scenario, build
# build a profiling variant next to the shipped lib
lib = os.path.join("gpurun_out", "libpgd_prof.so")
subprocess.check_call([build.hipcc(), '--offload-arch=gfx950',*build.OPT,'-std=c++17',*build.FAST_FP,'-shared','-fPIC','-DPGD_PROF','-o',lib, build.SRC] + os.environ.get('PGD_EXTRA','').split())
from pgdrive_amd import engine
engine._LIBH = None
L = engine.load_library(path=lib); engine._LIBH = L
L.pgd_debug_phase_cycles.argtypes=[C.c_void_p, C.c_void_p, C.c_int]
descs = bank.load_descriptions()
mb = mapdata.MapBank(descs); sb = scenario.ScenarioBank(descs,[d['seed'] for d in descs], traffic_mode=os.environ.get('TRAFFIC','trigger'))
N=int(sys.argv[2]) if len(sys.argv)>2 else 4096
mode = sys.argv[1] if len(sys.argv)>1 else 'uniform'
cfg=_abi.make_config(N)
eng = engine.Engine(cfg, mb, sb)
eng.reset(np.arange(N)%int(os.environ.get("NMAPS","100")))
rng=np.random.default_rng(0)
if mode in ('uniform','expert'):
    acts = torch.from_numpy(rng.uniform(-1,1,size=(64,N,1,2)).astype(np.float32)).cuda()
else:
    a = np.zeros((64,N,1,2),np.float32); a[...,1]=1.0; a[...,0]=rng.normal(0,0.05,size=(64,N,1)); acts=torch.from_numpy(a).cuda()
names=['load','trig+snap','policy','dynamics','crash','after_step','reward','reset','store','i_route','i_search','i_lc','i_pid','ld_stage','obs','WALL','as_route','as_getlane','as_local','as_side','o_pub','o_compact','o_state','o_neigh','o_lidar']
out=(C.c_ulonglong*64)()
abuf = torch.zeros((N,1,2),dtype=torch.float32,device='cuda')
def one(k):
    if mode=='expert' and k>0:
        eng.lane_keep_actions(abuf,k); eng.step(abuf)
    else: eng.step(acts[k%64])
with torch.cuda.stream(eng.stream):
    for k in range(1500): one(k)
    L.pgd_debug_phase_cycles(eng.h, out, 1)
    for rep in range(3):
        for k in range(300): one(1+k)
        L.pgd_debug_phase_cycles(eng.h, out, 1)
        tot=sum(out[:15])+sum(out[16:25]); nb=300*N
        print(mode, 'cycles/block:', {n:int(out[i]/nb) for i,n in enumerate(names)}, 'total', int(tot/nb), 'wall_ticks(100MHz)/block', round(out[15]/nb,1), '=> us', round(out[15]/nb/100,2), 'MHz', round(tot/max(out[15],1)*100))
        print('   MAX over blocks (cycles per step):', {n:int(out[32+i]/300) for i,n in enumerate(names)}, 'wall us', round(out[32+15]/300/100,2))
        f,i,ei=eng.get_state()
        print('   active traffic per env %.2f pending %.2f, ego kmh %.1f'%((i[0,:,1:]==2).sum(1).mean(), (i[0,:,1:]==1).sum(1).mean(), f[3,:,0].mean()*3.6))
