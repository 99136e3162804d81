#!/bin/bash
# kernel trace of the bench loop -> per-kernel stats, timeline summary and the text Gantt of one step.  OUT=gpurun_out/x bash tools/trace_step.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/${OUT:-gpurun_out/trace}
W=/tmp/trace_work
rm -rf $W && mkdir -p $O $W
rocprofv3 --kernel-trace --stats -d $W/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-meter ${BENCH_ARGS} > $O/bench_traced.json 2> $W/trace.err
cd $R
STEPS=$(python3 -c "import sqlite3;print(sqlite3.connect('$W/trace/bench_results.db').execute(\"select count(*) from kernels where name like '%adam_kernel%'\").fetchone()[0]//4)")
python3 tools/prof_summary.py $W/trace/bench_results.db $STEPS > $O/kernel_stats.txt 2>/dev/null
python3 tools/timeline.py $W/trace/bench_results.db > $O/timeline.txt 2>&1
python3 tools/step_trace.py $W/trace/bench_results.db 0 1 > $O/step_gantt.txt 2>&1
tail -1 $O/bench_traced.json | cut -c1-250
