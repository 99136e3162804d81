#!/bin/bash
# The world-1 RCCL rehearsal child (tests/dist_worker.py rccl1: backend nccl, CPCSV_FORCE_EXCHANGE=1, 8 steps behind the chunked
# asynchronous exchange, every piece captured) started N times as a FRESH process under torch.distributed.run; one line per run
# with its exit code, and the tail of the child's log for every run that did not exit 0. Round 4 saw 1 abort (SIGABRT from the
# ProcessGroupNCCL watchdog at teardown) in 14 such runs.
#   bash tools/rccl_soak.sh 30            # the current teardown (cpcsv.dist.shutdown)
#   OLD=1 bash tools/rccl_soak.sh 15      # round 4's teardown order (barrier; destroy), to catch its abort with the log kept
#   TRACE=1 ...                           # preload tools/abrt/libabrt_trace.so: native backtrace + thread name of whoever calls abort()
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
N=${1:-30}
OUT=${SOAK_OUT:-/tmp/rccl_soak}
mkdir -p $OUT
PRE=""
if [ "${TRACE:-0}" = "1" ]; then
  [ -f tools/abrt/libabrt_trace.so ] || gcc -O1 -g -shared -fPIC -o tools/abrt/libabrt_trace.so tools/abrt/abrt_trace.c
  PRE=$R/tools/abrt/libabrt_trace.so
fi
bad=0
echo "# rccl1 child x $N, CPCSV_OLD_TEARDOWN=${OLD:-0}"
for i in $(seq 1 $N); do
  port=$((29700 + i))
  LD_PRELOAD=$PRE CPCSV_OLD_TEARDOWN=${OLD:-0} CPCSV_FORCE_EXCHANGE=1 CPCSV_FUSED_MIN_NUMEL=256 HSA_ENABLE_IPC_MODE_LEGACY=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 \
    timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port \
    tests/dist_worker.py rccl1 $OUT/run_$i.npz > $OUT/run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then
    bad=$((bad + 1))
    echo "----- log tail of run $i -----"
    grep -v 'torch/distributed/\|^    \|^  File' $OUT/run_$i.log | head -150
    echo "------------------------------"
  fi
done
echo "# $bad of $N runs did not exit 0"
