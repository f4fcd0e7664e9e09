# round 5, call d: round 4's min-eigenvalue launch with SIX source rows in flight per lane (a0 .. a5) instead of three;
# staged-input leg on the `value` workload; first-steps leg; the default bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/d_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/d_tests.log
grep -E "^FAILED|^ERROR|Error|assert " gpurun_out/d_tests.log | head -30
run() {  # lib, label, extra env
env KVFE_LIB=$L/$1 $3 timeout 300 python bench.py --legs none --frames-persist --steps 52 --warmup 10 --repeats 3 --stage-event-stride 2 > gpurun_out/d_line.json 2> gpurun_out/d_err.log
python - "$1 $2" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:10], v) for k, v in st.items()))
PY
grep KVFE_ME_PROF gpurun_out/d_err.log
}
run libkvfe_base.so base ""
run libkvfe.so auto ""
for R in 120 96 60 48; do run libkvfe.so rows$R "KVFE_ME_ROWS=$R"; done
run libkvfe_base.so base ""
run libkvfe.so auto ""
run libkvfe_meprof.so prof-auto ""
echo "--- kf_realistic / c5 / single stream / nominal: base vs new"
for lib in libkvfe_base.so libkvfe.so; do
KVFE_LIB=$L/$lib timeout 300 python bench.py --legs kf_realistic,c5,single_stream,nominal --frames-persist --steps 52 --warmup 10 --repeats 2 --stage-event-stride 2 > gpurun_out/d_line.json 2> gpurun_out/d_err.log
python - $lib <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json'))
for k in ('kf_realistic','c5','single_stream','nominal'):
    v=d.get(k,{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
    print(sys.argv[1], k, v.get('value'), 'mineig %.3f' % st.get('mineig_localmax',0))
PY
done
echo "--- default bench line"
timeout 900 python bench.py > gpurun_out/d_bench_line.json 2> gpurun_out/d_bench.err; echo "bench rc=$? line bytes $(wc -c < gpurun_out/d_bench_line.json)"
cat gpurun_out/d_bench_line.json
cp bench_detail.json gpurun_out/d_bench_detail.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/d_bench_detail.json'))
print('pcie', d.get('pcie_inclusive')); print('first_steps', d.get('first_steps')); print('stages', d.get('stage_ms_per_step_summed_over_groups'))
PY
