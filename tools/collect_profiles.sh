#!/bin/bash
# Round evidence: (1) kernel trace + stats of the bench command, (2) FETCH_SIZE and WRITE_SIZE PMC passes (separate
# runs, no trace domains mixed in), (3) the default bench line. Run on the GPU box from the repo root:
#   bash tools/collect_profiles.sh        -> gpurun_out/r01/...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r01
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_traced.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-meter > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-meter > /dev/null 2> $O/pmc_write.err
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-300
ls -la $O $O/trace | head -20
