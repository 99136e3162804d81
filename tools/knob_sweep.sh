#!/bin/bash
# A/B of environment knobs on the bench step: one line per setting (ms per step, 20 timed steps each).
#   bash tools/knob_sweep.sh "CPCSV_WG_BLOCKS=256" "CPCSV_WG_BLOCKS=1024" ...      (first line = defaults)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo -n "$1: "; env $1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-meter 2>/dev/null | python3 -c "import sys,json; print([json.loads(l)['ms_per_step'] for l in sys.stdin if l.startswith('{')][-1])"; }
run "CPCSV_NOP=0"
for kv in "$@"; do run "$kv"; done
run "CPCSV_NOP=1"
