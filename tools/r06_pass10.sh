#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p10}
O=$R/gpurun_out/$TAG; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -q --timeout 600 -x -k "mlp_policy or throughput_mode" ) > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
timeout 600 python tools/mode_diff.py 3072 120 2>&1 | grep -v amdgpu.ids | tee $O/mode_diff.txt
timeout 300 python tools/mlp_bench.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/mlp_bench.txt
( timeout 900 python bench.py --no-cpu-baseline --rows c3_policy > $O/bench_rows.json 2> $O/bench_rows.err < /dev/null ) ; python - <<PY
import json
d=json.loads(open("$O/bench_rows.json").read().strip().splitlines()[-1])
r=[x for x in d["rows"] if x["row"]=="c3_policy"][0]
for impl in ("torch","fused"):
    for k,v in r.get(impl,{}).items():
        print(impl,k, {kk:(round(vv,2) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","us_per_iteration","host_enqueue_us_per_step","error","window_spread")} if isinstance(v,dict) else v)
print(r["value"], r["value_variant"])
PY
tail -3 $O/bench_rows.err
