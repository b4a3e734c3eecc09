"""Summarise the separate rocprofv3 --pmc passes of tools/profile_pass.sh into the traffic record bench.py reads
(profiles/rNN_pmc_traffic.json): HBM bytes per k_step launch = FETCH_SIZE * c_f + WRITE_SIZE * c_w, both corrections measured on
this engine's own access pattern with profiles/r01_calib.hip (the guide's gfx950 rule: FETCH_SIZE reports half of a wide
coalesced read)."""
import collections
import csv
import glob
import json
import sys

O = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096            # envs of the profiled run
ACTIONS = sys.argv[3] if len(sys.argv) > 3 else "uniform"      # bench.py --actions
MODE = sys.argv[4] if len(sys.argv) > 4 else "trigger"         # bench.py --traffic-mode
WORKLOAD = sys.argv[5] if len(sys.argv) > 5 else "c3"          # bench.py --workload
AGENTS = int(sys.argv[6]) if len(sys.argv) > 6 else 1          # bench.py --agents (c5)
LASERS = int(sys.argv[7]) if len(sys.argv) > 7 else (240 if WORKLOAD == "c3" else 72)
TRAFFIC = int(sys.argv[8]) if len(sys.argv) > 8 else (16 if WORKLOAD == "c3" else 0)  # bench.py --traffic


def pmc(d, steady_only=False):
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (O, d), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return acc


fetch, write, cf, cw = pmc("fetch"), pmc("write"), pmc("cal_fetch"), pmc("cal_write")
def most_dispatched(names, table):  # the instantiation the steps launch (pgd_reset launches another one once)
    return sorted(names, key=lambda k: -len(table[k]))


kname = most_dispatched([k for k in fetch if "k_step" in k], fetch)[0]
steady = lambda v: v[len(v) * 3 // 4:]  # the last quarter of the dispatches: steady-state traffic (after the 1500-step pre-roll)
fk = sum(steady(fetch[kname])) / len(steady(fetch[kname]))
wk = sum(steady(write[kname])) / len(steady(write[kname]))
rec_bytes, row_bytes = 4096 * 17 * 128, 4096 * 274 * 4
rc_f = sum(cf["rec_copy"]) / len(cf["rec_copy"])
rc_w = sum(cw["rec_copy"]) / len(cw["rec_copy"])
rw_w = sum(cw["row_write"]) / len(cw["row_write"])
c_f, c_w = rec_bytes / (rc_f * 1024.0), rec_bytes / (rc_w * 1024.0)
out = dict(envs=N, traffic=TRAFFIC, lasers=LASERS, actions=ACTIONS, traffic_mode=MODE, workload=WORKLOAD,
           agents=AGENTS if WORKLOAD == "c5" else 1, kernel=kname, FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk,
           dispatches_averaged=len(steady(fetch[kname])),
           calibration=dict(rec_copy_bytes=rec_bytes, rec_copy_FETCH_SIZE_KB=rc_f, rec_copy_WRITE_SIZE_KB=rc_w, row_write_bytes=row_bytes,
                            row_write_WRITE_SIZE_KB=rw_w, fetch_correction=c_f, write_correction_records=c_w,
                            write_correction_rows=row_bytes / (rw_w * 1024.0)),
           bytes_per_launch=(fk * c_f + wk * c_w) * 1024.0, bytes_per_env_step=(fk * c_f + wk * c_w) * 1024.0 / N,
           note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --exact --steps 200 --warmup 1500` "
                "(KB per dispatch, last quarter of the k_step dispatches = steady state). Corrections as MI355X_MICROARCH.md (HBM "
                "section) prescribes, measured on the engine's own record pattern with profiles/r01_calib.hip.")
# engines whose observation is a kernel of its own (multi-agent, many slots): the same figure for k_observe_env
ko = most_dispatched([k for k in fetch if "k_observe_env" in k and len(fetch[k]) >= 10], fetch)
if ko and ko[0] in write:
    fo = sum(steady(fetch[ko[0]])) / len(steady(fetch[ko[0]]))
    wo = sum(steady(write[ko[0]])) / len(steady(write[ko[0]]))
    out.update(k_observe_kernel=ko[0], k_observe_FETCH_SIZE_KB=fo, k_observe_WRITE_SIZE_KB=wo,
               bytes_per_launch_k_observe=(fo * c_f + wo * out["calibration"]["write_correction_rows"]) * 1024.0)
# the binary these passes ran on (bench.py quotes the record only for a library with the same stamp)
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pgdrive_amd import build as _build  # noqa: E402
out["source_sha"] = os.environ.get("PGD_PROFILE_SHA") or _build.source_sha()
print(json.dumps(out, indent=1))
