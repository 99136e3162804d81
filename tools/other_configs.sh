#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { line=$(python3 bench.py "$@" --steps 30 --warmup 10 --no-cpu-baseline --no-meter --no-trace --no-fp32-line 2>/dev/null | tail -1)
  echo "$* $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"; }
run --seq; run --st 32; run --cascade --st 32; run --clevr; run --cascade; run --graph
