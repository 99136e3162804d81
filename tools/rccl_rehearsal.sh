#!/bin/bash
# What the data-parallel machinery costs per step at the benchmark's widths, on ONE GPU: bench.py as rank 0 of a world-1 RCCL
# job with CPCSV_FORCE_EXCHANGE=1 (process group up, chunked asynchronous all-reduces between the graph pieces, optimiser steps
# behind the exchange instead of inside the backward graphs) against the plain single-GPU line. xGMI transfer time is NOT in
# this figure (one rank: the all-reduce is a local copy); it is the exposed launch / ordering / deferred-update cost.
#   bash tools/rccl_rehearsal.sh > profiles/r04_rccl_rehearsal.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "# bench.py --steps 30 --warmup 10, ST=12/IM=60 bf16, one MI355X"
for force in 0 1; do
  line=$(CPCSV_FORCE_EXCHANGE=$force python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus 1 --steps 30 --warmup 10 --no-cpu-baseline --no-meter --child 2>/dev/null | tail -1)
  echo "CPCSV_FORCE_EXCHANGE=$force  $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "story-frames/s", d["value"])')"
done
