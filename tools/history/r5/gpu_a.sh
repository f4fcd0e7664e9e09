# round 5, call a: min-eigenvalue launch with the per-stream mask bitmap + pulled work items (mineig_prep_kernel) against
# the round-4 build, strip-height sweep, wave balance of the new launch, the start-up transient probe, the new bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/a_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/a_tests.log
grep -E "Error|FAILED|assert " gpurun_out/a_tests.log | head -20
run() {  # lib, label, extra env
env KVFE_LIB=$L/$1 $3 timeout 300 python bench.py --legs none --frames-persist --steps 30 --warmup 8 --repeats 3 --stage-event-stride 2 > gpurun_out/a_line.json 2> gpurun_out/a_err.log
python - "$1 $2" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:10], v) for k, v in st.items()))
PY
grep KVFE_ME_PROF gpurun_out/a_err.log
}
run libkvfe_base.so base ""
run libkvfe.so auto ""
for R in 80 60 48 40 32; do run libkvfe.so rows$R "KVFE_ME_ROWS=$R"; done
run libkvfe_base.so base ""
run libkvfe.so auto ""
run libkvfe_meprof.so prof-auto ""
run libkvfe_meprof.so prof-80 "KVFE_ME_ROWS=80"
echo "--- kf_realistic / c5 / single stream: base vs new"
for lib in libkvfe_base.so libkvfe.so; do
KVFE_LIB=$L/$lib timeout 300 python bench.py --legs kf_realistic,c5,single_stream --frames-persist --steps 30 --warmup 8 --repeats 2 --stage-event-stride 2 > gpurun_out/a_line.json 2> gpurun_out/a_err.log
python - $lib <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json'))
for k in ('kf_realistic','c5','single_stream'):
    v=d.get(k,{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
    print(sys.argv[1], k, v.get('value'), 'mineig %.3f' % st.get('mineig_localmax',0))
PY
done
echo "--- transient probe"
timeout 300 python tools/r5/transient_probe.py 2>&1 | tail -20
echo "--- default bench line"
timeout 600 python bench.py > gpurun_out/a_bench_line.json 2> gpurun_out/a_bench.err; echo "bench rc=$? line bytes $(wc -c < gpurun_out/a_bench_line.json)"
cat gpurun_out/a_bench_line.json
cp bench_detail.json gpurun_out/a_bench_detail.json
