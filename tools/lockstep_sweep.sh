cd $GRAFT_REPO_ROOT
O=gpurun_out/t6_lockstep_probe.txt
: > $O
run() { env "$@" timeout 300 python tools/lockstep_probe.py 2>&1 | grep "step" >> $O; }
run X=1
run CPCSV_TEXT_FUSED=0
run CPCSV_STREAMS=0
run CPCSV_POISON=1
run SYNC=gfwd_before
run SYNC=gfwd_after
run SYNC=critic_after
run SYNC=critic_before
run SYNC=nograd_after
run PYTORCH_NO_CUDA_MEMORY_CACHING=1
run X=2
