#!/bin/bash
# Wall-clock share of each kernel family: bench.py with that family's launches skipped (numerically wrong, timing only).
# Kernel-time sums (rocprofv3) overstate what a family costs because the step runs on 5 streams; this measures it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${RND:-r04}/ablate.txt
echo "# bench.py --steps 20 --no-meter with one kernel family skipped (CPCSV_ABLATE); ms/step" > $O
run() { echo -n "$1: " >> $O; CPCSV_ABLATE="$2" CPCSV_BENCH_ALLOW_NONFINITE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-meter --child 2>/dev/null | python3 -c "import sys,json; print([json.loads(l)['ms_per_step'] for l in sys.stdin if l.startswith('{')][-1])" >> $O 2>&1; }
run baseline ""
run no_pack "cpcsv_pack_weight,cpcsv_pack_weight_sum"
run no_unpack "cpcsv_unpack_wgrad,cpcsv_unpack_wgrad_sum"
run no_adam "cpcsv_adam_step"
run no_spectral "cpcsv_spectral_sigma,cpcsv_spectral_sigma_multi"
run no_bn_bwd "cpcsv_bn_bwd_reduce,cpcsv_bn_bwd_apply"
run no_bn_fwd "cpcsv_bn_finalize,cpcsv_bn_apply,cpcsv_bn_apply_partials"
run no_wgrad "cpcsv_wgrad_tn,cpcsv_thin3x3_wgrad"
run no_gemm_nt "cpcsv_gemm_nt"
run no_pack_unpack_adam_sn "cpcsv_pack_weight,cpcsv_pack_weight_sum,cpcsv_unpack_wgrad,cpcsv_unpack_wgrad_sum,cpcsv_adam_step,cpcsv_spectral_sigma,cpcsv_spectral_sigma_multi"
cat $O
