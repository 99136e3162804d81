cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04mid; W=/tmp/mid_work; rm -rf $W; mkdir -p $O $W
rocprofv3 --kernel-trace --stats -d $W/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-meter --child > $O/bench_traced.json 2> $W/trace.err
cd $R
STEPS=$(python3 -c "import sqlite3;print(sqlite3.connect('$W/trace/bench_results.db').execute(\"select count(*) from kernels where name like '%adam_kernel%'\").fetchone()[0]//4)")
python3 tools/prof_summary.py $W/trace/bench_results.db $STEPS > $O/kernel_stats.txt 2>/dev/null
python3 tools/critical_path.py $W/trace/bench_results.db 1 8 > $O/critical_path.txt 2>&1
python3 tools/phase_times.py > $O/phase_times.txt 2>&1
