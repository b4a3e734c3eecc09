"""What the integer / flag differences of the respawn stream of tests/parity_campaign.py are (teacher-forced, one step each)."""
import sys, json; sys.path.insert(0,'.')
import numpy as np, torch
from tests import util
from oracle import orc
from pgdrive_amd import _abi, bank
from pgdrive_amd.engine import Engine
descs = bank.load_descriptions()
idm_agent = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
n_envs=1024
mb, sb = util.make_banks(descs, n_maps=100, traffic_mode="respawn")
cfg=_abi.make_config(n_envs, num_agents=1, num_traffic=16, num_lasers=240, auto_reset=1, seed=11, idm_agent=idm_agent)
eng=Engine(cfg,mb,sb); ora=orc.Oracle(cfg,mb,sb)
ids=np.arange(n_envs)%100
ora.reset(ids); eng.reset(ids)
rng=np.random.default_rng(17)
names={v:k for k,v in _abi.SI.items()}; fn={v:k for k,v in _abi.SF.items()}
n_int=n_flag=0; done=0
for t in range(steps):
    act=util.driving_actions(rng,n_envs)
    f0,i0,e0=ora.get_state()
    oo,orw,od,ofl=ora.step(act,threads=16)
    go,grw,gd,gfl=eng.step(torch.from_numpy(act).cuda()); eng.sync()
    gd=gd.cpu().numpy(); gfl=gfl.cpu().numpy().astype(np.uint32); done+=int(od.sum())
    bad=np.argwhere((gfl!=ofl)|(gd!=od))
    for e,a in bad:
        n_flag+=1; print("FLAG t=%d env=%d gpu=%#x orc=%#x done %d/%d"%(t,e,gfl[e,a],ofl[e,a],gd[e,a],od[e,a]))
    f,i,ei=ora.get_state(); gf,gi,gei=eng.get_state()
    for e,a in bad:
        # separating-axis gap between the ego's box and every other body, from the poses BEFORE the step's reset (f0 -> one step: use oracle's post poses if not reset)
        def box(F,s):
            sp=sb.spawn_table()[int(e0[_abi.EI['SCEN'] if hasattr(_abi,'EI') and 'SCEN' in _abi.EI else 0,e])] if False else None
            return F[0,e,s],F[1,e,s],F[24,e,s],F[25,e,s]
        for s_ in range(1,i.shape[2]):
            if i[0,e,s_]!=2: continue
            x0,y0,hx0,hy0=box(f,0); x1,y1,hx1,hy1=box(f,s_)
            print("   ego(orc post) %.6f %.6f h %.6f %.6f | slot %d %.6f %.6f h %.6f %.6f  d=%.6f"%(x0,y0,hx0,hy0,s_,x1,y1,hx1,hy1,np.hypot(x1-x0,y1-y0)))
    for k in range(i.shape[0]):
        for e,s in np.argwhere(gi[k]!=i[k]):
            n_int+=1
            print("INT t=%d env=%d slot=%d field=%s gpu=%d orc=%d  before: status=%d lane=%d rlane=%d timer=%d | x,y,v before=%.4f,%.4f,%.4f reset=%d"%(
                t,e,s,names[k],gi[k,e,s],i[k,e,s],i0[0,e,s],i0[1,e,s],i0[4,e,s],i0[5,e,s],f0[0,e,s],f0[1,e,s],f0[3,e,s], int(ofl[e,0]>>16&1)))
            print("     float diffs:", {fn[q]: (float(gf[q,e,s]), float(f[q,e,s])) for q in range(gf.shape[0]) if abs(float(gf[q,e,s])-float(f[q,e,s]))>1e-3})
    for e in np.argwhere((gei!=ei).any(axis=0)).ravel():
        print("EI t=%d env=%d"%(t,e), gei[:,e].tolist(), ei[:,e].tolist())
    f32=util.round_state_f32(f); ora.set_state(f32,i,ei); eng.set_state(f32,i,ei)
print(json.dumps(dict(idm_agent=idm_agent, steps=steps*n_envs, int=n_int, flag=n_flag, done=done)))
