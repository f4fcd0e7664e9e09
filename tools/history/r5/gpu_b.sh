# round 5, call b: min-eigenvalue launch, second form (per-stream counters, waves allotted to streams by cost); Harris GPU
# tests; what the periodic slow region of the headline loop is (stall_probe.py)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/b_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/b_tests.log
grep -E "Error|FAILED|assert " gpurun_out/b_tests.log | head -20
run() {  # lib, label, extra env
env KVFE_LIB=$L/$1 $3 timeout 300 python bench.py --legs none --frames-persist --steps 50 --warmup 10 --repeats 3 --stage-event-stride 2 > gpurun_out/b_line.json 2> gpurun_out/b_err.log
python - "$1 $2" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:10], v) for k, v in st.items()))
PY
grep KVFE_ME_PROF gpurun_out/b_err.log
}
run libkvfe_base.so base ""
run libkvfe.so auto ""
for R in 80 60 40 32; do run libkvfe.so rows$R "KVFE_ME_ROWS=$R"; done
run libkvfe_base.so base ""
run libkvfe.so auto ""
run libkvfe_meprof.so prof-auto ""
run libkvfe_meprof.so prof-80 "KVFE_ME_ROWS=80"
echo "--- kf_realistic / c5 / single stream: base vs new"
for lib in libkvfe_base.so libkvfe.so; do
KVFE_LIB=$L/$lib timeout 300 python bench.py --legs kf_realistic,c5,single_stream,nominal --frames-persist --steps 50 --warmup 10 --repeats 2 --stage-event-stride 2 > gpurun_out/b_line.json 2> gpurun_out/b_err.log
python - $lib <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json'))
for k in ('kf_realistic','c5','single_stream','nominal'):
    v=d.get(k,{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
    print(sys.argv[1], k, v.get('value'), 'mineig %.3f' % st.get('mineig_localmax',0))
PY
done
echo "--- stall probe"
PROBE_TAG=default timeout 200 python tools/r5/stall_probe.py 2>&1 | grep -v amdgpu.ids
PROBE_TAG=max_age_1000 PROBE_MAX_AGE=1000 timeout 200 python tools/r5/stall_probe.py 2>&1 | grep -v amdgpu.ids | head -2
PROBE_TAG=max_age_13 PROBE_MAX_AGE=13 timeout 200 python tools/r5/stall_probe.py 2>&1 | grep -v amdgpu.ids | head -2
