# round 4, call ad: mineig2_kernel with two waves per block (mask phase shared, strips of half the height); libkvfe_base.so = one wave
mkdir -p gpurun_out; export TMPDIR=/tmp
# (result: 109 parity tests green; min-eigenvalue launch 0.080 -> 0.087 ms, real frames 0.089 -> 0.110, 1280x720 0.113 -> 0.154: not kept,
# the two-wave kernel is not in the tree)
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py tests/test_gpu_components_r2.py tests/test_gpu_fuzz_slices.py -m gpu -q -x -k "not dense" > gpurun_out/ad_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/ad_tests.log | cut -c1-300
for V in libkvfe_base.so libkvfe.so libkvfe_base.so libkvfe.so; do
KVFE_LIB=$L/$V timeout 300 python bench.py --legs kf_realistic,c5 --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/ad_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$V]', d['value'], d['ms_per_step'], d['repeats']['values'], 'mineig %.4f' % st['mineig_localmax'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'])
for k in ('kf_realistic','c5'):
    print('   ', k, d[k]['value'], 'mineig %.4f' % d[k]['stage_ms_per_step_summed_over_groups']['mineig_localmax'])
"
done
