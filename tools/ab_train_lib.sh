#!/bin/bash
# Same-box A/B of the train step between the default library and the -DGT_EXP build (GTCRN_LIB_VARIANT=exp), alternating
# child processes (a process keeps its physical memory placement: single runs of one binary differ by up to 3 %).
#   tools/ab_train_lib.sh [storage=f32] [pairs=3] [mask=65535]
ST=${1:-f32}; N=${2:-3}; M=${3:-65535}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in $(seq 1 $N); do
  for v in exp base; do
    if [ $v = exp ]; then export GTCRN_LIB_VARIANT=exp; else unset GTCRN_LIB_VARIANT; fi
    echo -n "$v "; python3 "$R/tools/ab_train_fusions.py" --storage $ST --masks $M --rounds 2 2>&1 | tail -1
  done
done
