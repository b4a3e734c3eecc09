"""Summarise the SQ instruction-counter passes of tools/row_pass.sh into the record bench.py reads for `roofline.issue`
(profiles/rNN_<row>_pmc_insts.json): instructions per wave of k_step (and k_observe_env) by class, waves per launch, and the
busy / wait shares of a wave's life (SQ_* cycle counters count quad-cycles, MI355X_MICROARCH.md:446).
usage: pmc_insts.py DIR ENVS ACTIONS MODE WORKLOAD AGENTS LASERS      (DIR holds insts_a/, insts_b/, insts_c/ from rocprofv3 --pmc)"""
import collections
import csv
import glob
import json
import sys

O = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ACTIONS = sys.argv[3] if len(sys.argv) > 3 else "uniform"
MODE = sys.argv[4] if len(sys.argv) > 4 else "trigger"
WORKLOAD = sys.argv[5] if len(sys.argv) > 5 else "c3"
AGENTS = int(sys.argv[6]) if len(sys.argv) > 6 else 1
LASERS = int(sys.argv[7]) if len(sys.argv) > 7 else (240 if WORKLOAD == "c3" else 72)
TRAFFIC = int(sys.argv[8]) if len(sys.argv) > 8 else (16 if WORKLOAD == "c3" else 0)

acc = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> per-dispatch values
for f in glob.glob("%s/insts_*/**/*counter_collection.csv" % O, recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))


def steady(v):
    return v[len(v) * 3 // 4:]  # the last quarter of the dispatches: steady state (after the pre-roll)


def mean(v):
    v = steady(v)
    return sum(v) / len(v) if v else None


def summarise(kname):
    c = {k: mean(v) for k, v in acc[kname].items()}
    waves = c.get("SQ_WAVES")
    if not waves:
        return None
    per = lambda k: (c[k] / waves) if c.get(k) is not None else None  # noqa: E731
    valu, salu, lds = per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS")
    vmem = (per("SQ_INSTS_VMEM_RD") or 0) + (per("SQ_INSTS_VMEM_WR") or 0)
    smem, branch = per("SQ_INSTS_SMEM"), per("SQ_INSTS_BRANCH")
    total = sum(x for x in (valu, salu, lds, vmem, smem) if x)
    out = dict(kernel=kname, waves_per_launch=waves, insts_per_wave=total, valu=valu, salu=salu, lds=lds, vmem=vmem, smem=smem,
               branch_of_salu=branch, dispatches_averaged=len(steady(acc[kname]["SQ_WAVES"])))
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        out.update(wave_cycles_per_wave=4 * wc / waves,  # quad-cycles -> cycles
                   wait_any_share=(c.get("SQ_WAIT_ANY") or 0) / wc if c.get("SQ_WAIT_ANY") is not None else None,
                   wait_inst_any_share=(c.get("SQ_WAIT_INST_ANY") or 0) / wc if c.get("SQ_WAIT_INST_ANY") is not None else None,
                   active_inst_any_share=(c.get("SQ_ACTIVE_INST_ANY") or 0) / wc if c.get("SQ_ACTIVE_INST_ANY") is not None else None,
                   busy_cycles=c.get("SQ_BUSY_CYCLES"))
    return out


def by_dispatches(names):  # the instantiation the steps launch (pgd_reset launches another one once)
    names = [k for k in names if len(acc[k].get("SQ_WAVES", [])) >= 10]
    return sorted(names, key=lambda k: -len(acc[k]["SQ_WAVES"]))


ks = by_dispatches([k for k in acc if "k_step" in k])
ko = by_dispatches([k for k in acc if "k_observe_env" in k])
out = dict(envs=N, traffic=TRAFFIC, lasers=LASERS, actions=ACTIONS, traffic_mode=MODE, workload=WORKLOAD,
           agents=AGENTS if WORKLOAD == "c5" else 1,
           k_step=summarise(ks[0]) if ks else None, k_observe=summarise(ko[0]) if ko else None,
           note="rocprofv3 --pmc passes (SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES | SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM "
                "SQ_INSTS_BRANCH | SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY) over `python bench.py --exact --steps 200 "
                "--warmup 1500` of the row; per-dispatch sums, last quarter of the dispatches, divided by SQ_WAVES")
# the binary these passes ran on (bench.py quotes the record only for a library with the same stamp)
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pgdrive_amd import build as _build  # noqa: E402
out["source_sha"] = os.environ.get("PGD_PROFILE_SHA") or _build.source_sha()
print(json.dumps(out, indent=1))
