# round 6: per-kernel durations of the LK stage (eight-point waves + deferred one-point pass) per iteration cap
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
cd /tmp
for CAP in ${CAPS:-4 6 30}; do
KVFE_LK_IMPL=0 KVFE_LK8_CAP=$CAP timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lk8t_$CAP -o s -- python $R/bench.py --steps 12 --warmup 4 --repeats 1 --legs none --no-stage-events --no-cpu-baseline > $R/gpurun_out/lk8t_$CAP.log 2>&1; echo "cap $CAP rc=$?"
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/lk8t_$CAP -name "*.db" | head -1) 2>&1 | grep -i "lk8_kernel\|lk_kernel\|^| kernel" | head -4
done
rm -rf $R/gpurun_out/lk8t_*/
