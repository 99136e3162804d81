#!/bin/bash
# What the data-parallel machinery costs per step at the benchmark's widths, on ONE GPU: bench.py as rank 0 of a world-1 RCCL
# job with CPCSV_FORCE_EXCHANGE=1 (process group up, every collective on the one communication stream between the graph pieces,
# optimiser steps behind the exchange instead of inside the backward graphs) against the plain single-GPU line. xGMI transfer time
# is NOT in this figure (one rank: the all-reduce is a local copy); it is the exposed launch / ordering / deferred-update cost.
#   bash tools/rccl_rehearsal.sh > profiles/r06_rccl_rehearsal.txt
#   QUEUES="0 1 2 3" bash tools/rccl_rehearsal.sh      # which hardware queue the communication stream shares (CPCSV_COMM_QUEUE)
#   WIRES="bf16 fp32" ...                              # gradient payload (CPCSV_GRAD_COMM)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "# bench.py --steps 30 --warmup 10, ST=12/IM=60 bf16, one MI355X"
run() {
  line=$(env "$@" python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus 1 --steps 30 --warmup 10 --no-cpu-baseline --no-meter --child 2>/dev/null | tail -1)
  echo "$*  $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "story-frames/s", d["value"])')"
}
run CPCSV_FORCE_EXCHANGE=0
for q in ${QUEUES:-default}; do
  for w in ${WIRES:-bf16}; do
    if [ "$q" = default ]; then run CPCSV_FORCE_EXCHANGE=1 CPCSV_GRAD_COMM=$w       # (the trainer's own queue pairing: CPCSV_COMM_QUEUE unset)
    else run CPCSV_FORCE_EXCHANGE=1 CPCSV_COMM_QUEUE=$q CPCSV_GRAD_COMM=$w; fi
  done
done
run CPCSV_FORCE_EXCHANGE=0
