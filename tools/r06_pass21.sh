#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p21}
O=$R/gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do for nw in 4 2 8; do
  echo "--- k_observe_env with $nw waves per env"; PGD_KOE_NW=$nw bash tools/ab40.sh -r 1 koe 2>&1 | grep "agents 40"
done; done | tee $O/koe_nw.txt
