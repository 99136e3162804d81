#!/bin/bash
# Diagnostic: the world-1 RCCL rehearsal worker (tests/dist_worker.py rccl1) N times, full output kept.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=${1:-gpurun_out/rccl_try}
mkdir -p $O
for i in 1 2 3; do
  CPCSV_FORCE_EXCHANGE=1 CPCSV_FUSED_MIN_NUMEL=256 HSA_ENABLE_IPC_MODE_LEGACY=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $EXTRA_ENV \
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29700 + i)) \
    tests/dist_worker.py rccl1 /tmp/rccl_try_$i.npz > $O/try_$i.log 2>&1
  echo "try $i rc=$?"
  grep -m3 -E "HIP error|hipError|Error|error:" $O/try_$i.log | cut -c1-300
done
