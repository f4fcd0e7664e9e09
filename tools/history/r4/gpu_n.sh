# round 4, call n: fused C(p,d) kernel with the records staged through LDS one row ahead; 16 / 12 / 8 columns per wave
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py tests/test_gpu_components_r2.py tests/test_gpu_fuzz_slices.py -m gpu -q -x -k "dense or fuzz" > gpurun_out/n_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/n_tests.log | cut -c1-300
for V in libkvfe.so libkvfe_n12.so libkvfe_n8.so; do
KVFE_LIB=$L/$V timeout 300 python bench.py --legs dense,dense_c5 --steps 10 --warmup 4 --repeats 1 2> gpurun_out/n_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read())
for k in ('dense_stereo','dense_stereo_c5'):
    print('$V', k, {a: d[k].get(a) for a in ('value','ms_per_pair','ms_per_pair_min')})
"
done
KVFE_LIB=$L/libkvfe_n12.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dense" > gpurun_out/n_tests12.log 2>&1; echo "pytest n12 rc=$?"; tail -2 gpurun_out/n_tests12.log | cut -c1-300
cd /tmp
for V in libkvfe.so libkvfe_n12.so; do
KVFE_LIB=$L/$V timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_n_$V -o kt -- python $R/bench.py --steps 4 --warmup 2 --repeats 1 --legs dense > $R/gpurun_out/prof_n.log 2>&1; echo "kt rc=$?"
db=$(find $R/gpurun_out/prof_n_$V -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db | grep -E "dense|speckle" | cut -c1-150
done
