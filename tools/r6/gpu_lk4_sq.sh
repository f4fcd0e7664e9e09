# round 6: SQ counters of the LK launch (KVFE_LK_IMPL: 0 = four points per wave, 1 = one point per wave), one rocprofv3
# --pmc pass per counter group (never combined with other trace domains; TA_* / TCP_* groups hang on this pool)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
cd /tmp
pass() {
  impl=$1; n=$2; shift; shift
  KVFE_LK_IMPL=$impl timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/lk4c_${impl}_$n -o s -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --legs none --no-stage-events --no-cpu-baseline > $R/gpurun_out/lk4c_${impl}_$n.log 2>&1; echo "impl $impl pass $n rc=$?"
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/lk4c_${impl}_$n -name "*.db" | head -1) 2>&1 | grep -i "^| kernel\|lk4_kernel\|lk_kernel" | head -4
}
for impl in ${IMPLS:-0}; do
  pass $impl 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
  pass $impl 2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT
done
rm -rf $R/gpurun_out/lk4c_*/
