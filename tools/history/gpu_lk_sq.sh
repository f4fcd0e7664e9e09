# SQ / SQC counters of the batched leg for one or two builds of the library on one box (profiles/r2_v5_lk_analysis.md,
# table 2).  One rocprofv3 --pmc pass per counter group (never combined with other trace domains).
# NOTE: the TA_* and TCP_* counter groups hung rocprofv3 on this pool (two passes ran into their timeouts and cost ten
# GPU-minutes): they are left out on purpose.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; L=$R/kimera_vio_amd/csrc
cd /tmp
pass() {
  lib=$1; n=$2; shift; shift
  KVFE_LIB=$L/$lib.so timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/lkc_${lib}_$n -o s -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --legs none --no-stage-events > $R/gpurun_out/lkc_${lib}_$n.log 2>&1; echo "$lib pass $n rc=$?"
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/lkc_${lib}_$n -name "*.db" | head -1) 2>&1 | grep -i "^| kernel\|lk_kernel\|mineig\|stereo_match" | head -4
}
for lib in libkvfe_base libkvfe; do
  [ -f $L/$lib.so ] || continue
  pass $lib 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
  pass $lib 2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH
  pass $lib 3 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM
done
