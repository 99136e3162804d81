#!/bin/bash
# L2 behaviour of the GEMM kernels: one rocprofv3 --pmc pass (TCC hit / miss / request counters) over a few bench steps.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${RND:-r03}
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/pmc_l2_counters_available.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d /tmp/pmc_l2 -o bench -- python3 $R/bench.py --steps 3 --warmup 5 --no-cpu-baseline --no-meter > /dev/null 2> /tmp/pmc_l2.err
cd $R
python3 tools/pmc_generic.py /tmp/pmc_l2/bench_results.db 12 > $O/pmc_l2.txt 2>&1 || tail -5 /tmp/pmc_l2.err >> $O/pmc_l2.txt
cat $O/pmc_l2.txt
