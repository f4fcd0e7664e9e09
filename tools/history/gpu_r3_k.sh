# round 3, call k: select_kernel with the discs of a batch's accepted corners marked after the decision loop (G corners
# per trip): full suite, then libkvfe_base.so (previous build) against libkvfe.so on the select-heavy legs, and the
# kernel's own phase stamps for both on the single-stream workload.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; L=$PWD/kimera_vio_amd/csrc
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/k_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/k_tests.log
[ $rc -ne 0 ] && grep -E "Error|FAILED|assert" gpurun_out/k_tests.log | head -12
run() {
KVFE_LIB=$L/$1 timeout 300 python bench.py --legs single_stream,c5,kf_realistic --steps 20 --warmup 8 --repeats 2 2> gpurun_out/k_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], [(k, d[k]['value'], d[k]['stage_ms_per_step_summed_over_groups'].get('gftt_select') if 'stage_ms_per_step_summed_over_groups' in d[k] else None) for k in ('single_stream','c5','kf_realistic') if k in d])"
}
for rep in 1 2; do run libkvfe_base.so; run libkvfe.so; done
for lib in libkvfe_base.so libkvfe.so; do
KVFE_LIB=$L/$lib KVFE_SELECT_PROF=1 timeout 200 python bench.py --config c2 --steps 100 --warmup 10 --repeats 1 --legs none --no-stage-events > gpurun_out/k_sel.json 2> gpurun_out/k_sel.err; echo $lib; grep KVFE_SELECT_PROF gpurun_out/k_sel.err
done
