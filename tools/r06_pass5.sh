#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p5}
O=$R/gpurun_out/$TAG; mkdir -p $O
( time timeout 600 python -m pytest tests -m gpu -q --timeout 600 -x -k "mlp_policy or instantiations or test_marl_free_running or observation_kernel_equals or idm_agent_parity" ) > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
( timeout 900 python bench.py --no-cpu-baseline --rows c3_policy > $O/bench_rows.json 2> $O/bench_rows.err < /dev/null ) ; python - <<PY
import json
d=json.loads(open("$O/bench_rows.json").read().strip().splitlines()[-1])
r=[x for x in d["rows"] if x["row"]=="c3_policy"][0]
for impl in ("torch","fused"):
    for k,v in r.get(impl,{}).items():
        print(impl,k, {kk:(round(vv,2) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","us_per_iteration","host_enqueue_us_per_step","error","window_spread")} if isinstance(v,dict) else v)
PY
bash tools/ab3.sh -r 2 ckptchain new 2>&1 | tee $O/ab3.txt
echo "--- 40 / 8 seats: uniform body size on"; bash tools/ab40.sh -r 2 head 2>&1 | tee $O/ab40_uni.txt
echo "--- off (PGD_NO_UNI=1)"; PGD_NO_UNI=1 bash tools/ab40.sh -r 2 head 2>&1 | tee $O/ab40_nouni.txt
echo "--- ckptchain"; bash tools/ab40.sh -r 1 ckptchain 2>&1 | tee $O/ab40_chain.txt
