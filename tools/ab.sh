#!/bin/bash
# A/B runs of the bench under environment knobs, one child per setting:  bash tools/ab.sh "NAME=VAL ..." "NAME2=VAL2" ...
#   ("" = defaults). Prints ms/step of `bench.py --steps ${STEPS:-40} --warmup 10` (no meter, no CPU baseline, no children).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "$@"; do
  line=$(env $cfg python3 bench.py --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-meter --child ${BENCH_ARGS} 2>/dev/null | tail -1)
  ms=$(echo "$line" | python3 -c 'import json,sys
try:
    d=json.loads(sys.stdin.read()); print("%.3f ms/step  (G loss %s)" % (d["ms_per_step"], d["config"]["G_loss_after"]))
except Exception as e:
    print("FAILED", e)')
  echo "[${cfg:-defaults}] $ms"
done
