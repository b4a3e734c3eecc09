"""Per-wave life histogram of k_step (one env per wave: a wave = an env).  -DPGD_PROF build: every wave adds the shader-clock cycles
between its phase marks and its wall-clock life (100 MHz ticks) to its block's counters; a launch lasts as long as its slowest wave,
so slowest / mean over the waves of a step is what the tail costs.  Prints, over STEPS sampled steady-state steps: mean / p50 / p90 /
p99 / max of the wave life, slowest / mean per step (mean and worst step), the histogram, the mean life by class of env (restarting,
0 / 1-2 / 3-5 / 6+ driving IDM vehicles) and the phases of the slowest waves.
usage: [TRAFFIC=respawn] wave_life.py [uniform|straight|expert] [N] [STEPS]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: E402
from pgdrive_amd import _abi, bank, mapdata, scenario, build  # noqa: E402

lib = os.path.join("gpurun_out", "libpgd_prof.so")
os.makedirs("gpurun_out", exist_ok=True)
subprocess.check_call([build.hipcc(), '--offload-arch=gfx950', *build.OPT, '-std=c++17', *build.FAST_FP, '-shared', '-fPIC', '-DPGD_PROF',
                       '-o', lib, build.SRC] + os.environ.get('PGD_EXTRA', '').split())
from pgdrive_amd import engine  # noqa: E402
L = engine.load_library(path=lib)
engine._LIBH = L
mode = sys.argv[1] if len(sys.argv) > 1 else 'uniform'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 24
traffic = os.environ.get('TRAFFIC', 'trigger')
descs = bank.get_descriptions(range(1000, 1100))
mb = mapdata.MapBank(descs)
sb = scenario.ScenarioBank(descs, [d['seed'] for d in descs], traffic_mode=traffic)
eng = engine.Engine(_abi.make_config(N, auto_reset=1, seed=1234), mb, sb)
eng.reset(np.arange(N) % 100)
rng = np.random.default_rng(0)
if mode in ('uniform', 'expert'):
    acts = torch.from_numpy(rng.uniform(-1, 1, size=(64, N, 1, 2)).astype(np.float32)).cuda()
else:
    a = np.zeros((64, N, 1, 2), np.float32)
    a[..., 1] = 1.0
    a[..., 0] = rng.normal(0, 0.05, size=(64, N, 1))
    acts = torch.from_numpy(a).cuda()
names = ['load', 'trig+snap', 'policy', 'dynamics', 'crash', 'after_step', 'reward', 'reset', 'store', 'i_route', 'i_search', 'i_lc', 'i_pid',
         'ld_stage', 'obs', 'WALL', 'as_route', 'as_getlane', 'as_local', 'as_side', 'o_pub', 'o_compact', 'o_state', 'o_neigh', 'o_lidar',
         'as_vehicle']
raw = (C.c_ulonglong * (N * 32))()
L.pgd_debug_phase_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
abuf = torch.zeros((N, 1, 2), dtype=torch.float32, device='cuda')


def one(k):
    if mode == 'expert' and k > 0:
        eng.lane_keep_actions(abuf, k)
        return eng.step(abuf)
    return eng.step(acts[k % 64])


print('wave life of k_step: %d envs, actions %s, traffic mode %s; %s' % (N, mode, traffic, eng.describe_step() if hasattr(eng, 'describe_step') else ''))
lives, ratios, cls_acc = [], [], {}
slow_rows = []
with torch.cuda.stream(eng.stream):
    for k in range(1500):
        one(k)
    L.pgd_debug_phase_raw(eng.h, raw, N)
    k = 1500
    for smp in range(STEPS):
        for _ in range(7):  # the sampled steps are 8 apart
            one(k)
            k += 1
        L.pgd_debug_phase_raw(eng.h, raw, N)  # (clears the counters)
        f, i, ei = eng.get_state()
        nact = (i[_abi.SI["STATUS"]][:, 1:] == _abi.ST_ACTIVE).sum(1)
        o, r, d, fl = one(k)
        k += 1
        eng.sync()
        L.pgd_debug_phase_raw(eng.h, raw, N)
        a = np.frombuffer(raw, dtype=np.uint64).reshape(N, 32).astype(np.int64)
        cyc = a[:, :15].sum(1) + a[:, 16:32].sum(1)  # shader-clock cycles between the first and the last mark of the wave
        lives.append(cyc)
        ratios.append(cyc.max() / cyc.mean())
        dn = d.cpu().numpy().reshape(-1)
        for lab, m in (('restarting', dn == 1), ('0 driving', (nact == 0) & (dn == 0)), ('1-2 driving', (nact > 0) & (nact < 3) & (dn == 0)),
                       ('3-5 driving', (nact >= 3) & (nact < 6) & (dn == 0)), ('6+ driving', (nact >= 6) & (dn == 0))):
            if m.sum():
                c = cls_acc.setdefault(lab, [0, 0.0, 0.0])
                c[0] += int(m.sum())
                c[1] += float(cyc[m].sum())
                c[2] = max(c[2], float(cyc[m].max()))
        b = int(np.argmax(cyc))
        slow_rows.append((int(cyc[b]), int(nact[b]), int(dn[b]), {n: int(a[b, j]) for j, n in enumerate(names) if j != 15 and a[b, j] > 1500}))
allc = np.concatenate(lives).astype(np.float64)
print('driving IDM vehicles per env (last sample): %.2f' % nact.mean())
print('cycles per wave over %d steps x %d waves: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f' % (
    STEPS, N, allc.mean(), np.percentile(allc, 50), np.percentile(allc, 90), np.percentile(allc, 99), allc.max()))
print('slowest / mean per step: mean %.3f  worst %.3f  best %.3f     p99 / mean (all samples) %.3f' % (
    np.mean(ratios), np.max(ratios), np.min(ratios), np.percentile(allc, 99) / allc.mean()))
edges = np.arange(0, allc.max() + 2000, 2000)
hist, _ = np.histogram(allc, bins=edges)
print('histogram (2000-cycle bins; share of waves):')
for lo, h in zip(edges[:-1], hist):
    if h:
        print('  %6d - %6d  %7.3f %%  %s' % (lo, lo + 2000, 100.0 * h / allc.size, '#' * int(round(60.0 * h / hist.max()))))
print('mean life by class of env:')
for lab, c in cls_acc.items():
    print('  %-12s %6.2f %% of waves  mean %6.0f  max %6.0f' % (lab, 100.0 * c[0] / allc.size, c[1] / c[0], c[2]))
print('the slowest wave of each sampled step (cycles, driving IDM vehicles, restarting, phases above 1500 cycles):')
for row in sorted(slow_rows, key=lambda t: -t[0])[:12]:
    print('  %6d  nact %2d  done %d  %s' % row)
