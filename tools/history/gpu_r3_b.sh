mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 300 tools/ubench/valu_rate > gpurun_out/valu_rate2.txt 2>&1; echo "valu_rate rc=$?"; cut -c1-200 gpurun_out/valu_rate2.txt | sed -e 's/(wall@2.4GHz), //g' -e 's/cyc\/instr\/SIMD//g'
timeout 600 python -m pytest tests/test_gpu_pyramid_r3.py -x -q > gpurun_out/pyr_tests.log 2>&1; echo "pyr tests rc=$?"; tail -5 gpurun_out/pyr_tests.log
VARIANTS="KVFE_PYR_T2=2|KVFE_PYR_T2=4|KVFE_PYR_T2=8" REPEATS=1 bash tools/gpu_ab.sh
Q="--steps 10 --warmup 3 --repeats 1 --legs none --no-stage-events"
cd /tmp
for t in 4 8; do
KVFE_PYR_T2=$t timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3b_kt$t -o kt -- python $R/bench.py $Q > $R/gpurun_out/r3b_kt$t.log 2>&1; echo "kt rc=$?"
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/r3b_kt$t -name "*.db" | head -1) | grep -E "pyr2|kernel "
done
