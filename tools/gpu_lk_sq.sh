# what saturates while lk_kernel_sys runs: instruction cache, vector-memory path (TA / TCP / UTCL1), issue cycles by
# instruction class; batched leg alone, one rocprofv3 --pmc pass per counter group
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; L=$R/kimera_vio_amd/csrc
cd /tmp
pass() {
  n=$1; shift
  KVFE_LIB=$L/libkvfe.so timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/lkc_$n -o s -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --legs none --no-stage-events > $R/gpurun_out/lkc_$n.log 2>&1; echo "pass $n rc=$?"
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/lkc_$n -name "*.db" | head -1) 2>&1 | grep -i "^| kernel\|lk_kernel\|mineig\|stereo_match" | head -4
}
pass 1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH
pass 2 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM
pass 3 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN2_sum
pass 4 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
