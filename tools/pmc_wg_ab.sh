#!/bin/bash
# PMC read traffic of the weight-gradient kernel with and without the XCD-aware (tile, split) map
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for v in 1 0; do
  export CPCSV_WG_XCD=$v
  W=/tmp/pmcab_$v; rm -rf $W; mkdir -p $W
  rocprofv3 --pmc FETCH_SIZE -d $W/pmc_fetch -o bench -- python3 $R/bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-meter --child > /dev/null 2> $W/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE -d $W/pmc_write -o bench -- python3 $R/bench.py --steps 4 --warmup 5 --no-cpu-baseline --no-meter --child > /dev/null 2> $W/pmc_write.err
  (cd $R; python3 tools/pmc_summary.py $W/pmc_fetch/bench_results.db $W/pmc_write/bench_results.db $O/pmc_wgxcd_$v.json > $O/pmc_wgxcd_$v.txt 2>&1)
  echo "== CPCSV_WG_XCD=$v"; head -8 $O/pmc_wgxcd_$v.txt
done
