#!/bin/bash
# Round evidence. Run on the GPU box from the repo root:   RND=r04 SHA=<commit> bash tools/collect_profiles.sh
#   (1) kernel trace + stats of the bench command (+ timeline, text Gantt of one step, per-phase critical path),
#   (2) FETCH_SIZE and WRITE_SIZE PMC passes (separate runs, no trace domains mixed in), (3) MFMA-busy / wave-cycle PMC pass,
#   (4) the default bench line (with cpu_baseline; its roofline.traffic comes from pass (2) of THIS run), fp32 and cascade lines,
#   (5) GPU time per phase (events) and host enqueue time, (6) wall-clock share per kernel family (tools/ablate.sh).
# Writes gpurun_out/$RND/*.txt|json (the rocpd databases are summarised on the box; they are too big to travel back).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=${RND:-r04}
export SHA=${SHA:-unknown}
O=$R/gpurun_out/$RND
W=/tmp/${RND}_work
rm -rf $W && mkdir -p $O $W
echo "$SHA" > $O/HEAD
rocprofv3 --kernel-trace --stats -d $W/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-meter --child > $O/bench_traced.json 2> $W/trace.err
rocprofv3 --pmc FETCH_SIZE -d $W/pmc_fetch -o bench -- python3 $R/bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-meter --child > /dev/null 2> $W/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $W/pmc_write -o bench -- python3 $R/bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-meter --child > /dev/null 2> $W/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $W/pmc_mfma -o bench -- python3 $R/bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-meter --child > /dev/null 2> $W/pmc_mfma.err
cd $R
STEPS=$(python3 -c "import sqlite3;print(sqlite3.connect('$W/trace/bench_results.db').execute(\"select count(*) from kernels where name like '%adam_kernel%'\").fetchone()[0]//4)")
python3 tools/prof_summary.py $W/trace/bench_results.db $STEPS > $O/kernel_stats.txt 2>/dev/null
python3 tools/prof_by_grid.py $W/trace/bench_results.db gemm $STEPS > $O/gemm_by_grid.txt 2>/dev/null
python3 tools/pmc_summary.py $W/pmc_fetch/bench_results.db $W/pmc_write/bench_results.db $O/pmc_traffic.json > $O/pmc_traffic.txt
python3 tools/timeline.py $W/trace/bench_results.db > $O/timeline.txt 2>&1
python3 tools/step_trace.py $W/trace/bench_results.db 0 1 > $O/step_gantt.txt 2>&1
python3 tools/critical_path.py $W/trace/bench_results.db 1 8 > $O/critical_path.txt 2>&1
python3 tools/pmc_mfma.py $W/pmc_mfma/bench_results.db > $O/pmc_mfma.txt 2> $O/pmc_mfma.err || tail -3 $W/pmc_mfma.err >> $O/pmc_mfma.err
CPCSV_PMC_TRAFFIC_JSON=$O/pmc_traffic.json CPCSV_BENCH_SHAPES=$O/gemm_by_shape.txt python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --dtype fp32 --no-cpu-baseline --child > $O/bench_fp32.json 2> $O/bench_fp32.err
python3 bench.py --cascade --no-cpu-baseline --no-fp32-line > $O/bench_cascade.json 2> $O/bench_cascade.err
python3 tools/phase_times.py > $O/phase_times.txt 2>&1
HOST_PROFILE_SHORT=1 python3 tools/host_profile.py 2>&1 | tail -1 >> $O/phase_times.txt
RND=$RND bash tools/ablate.sh > /dev/null 2>&1
python3 tools/roofline_table.py $O > $O/roofline.txt 2>&1
tail -1 $O/bench_default.json | cut -c1-200
ls -la $O
