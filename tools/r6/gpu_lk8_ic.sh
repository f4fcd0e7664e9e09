# round 6: instruction-cache counters of the LK launch
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
cd /tmp
for impl in ${IMPLS:-0}; do
KVFE_LK_IMPL=$impl timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/lk8ic_$impl -o s -- python $R/bench.py --steps 6 --warmup 2 --repeats 1 --legs none --no-stage-events --no-cpu-baseline > $R/gpurun_out/lk8ic_$impl.log 2>&1; echo "impl $impl rc=$?"
python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/lk8ic_$impl -name "*.db" | head -1) 2>&1 | grep -i "^| kernel\|lk8_kernel\|lk_kernel" | head -4
done
rm -rf $R/gpurun_out/lk8ic_*/
