#!/bin/bash
# read traffic and duration per launch geometry of the GEMM family under the two K orders of the streaming main loop
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in 0 2; do
  export CPCSV_KORDER=$v
  rm -rf /tmp/pk_$v
  rocprofv3 --pmc FETCH_SIZE -d /tmp/pk_$v -o bench -- python3 $R/bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-meter --child > /dev/null 2> /tmp/pk_$v.err
  (cd $R; python3 tools/pmc_by_grid.py /tmp/pk_$v/bench_results.db gemm_nt > gpurun_out/pmc_korder_$v.txt 2>&1)
done
