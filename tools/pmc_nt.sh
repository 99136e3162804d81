cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=$R/tools/probe/nt_probe_0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr"; do
  tag=$(echo $grp | cut -d' ' -f1)
  CPCSV_NT_BIG=0 rocprofv3 --pmc $grp -d $R/gpurun_out/pmcnt/$tag -o p -- $P 8192 2048 8192 5 > /dev/null 2>$R/gpurun_out/pmcnt_$tag.err || echo "fail $tag"
done
rocprofv3 -L 2>/dev/null | grep -o "^\s*[A-Z][A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/counters.txt
ls $R/gpurun_out/pmcnt/*
