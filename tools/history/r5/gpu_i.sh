# round 5, call i: points past maxFeatureAge are not tracked (their result is dropped anyway); smaller grid of the
# one-corner cornerSubPix kernel; A/B against the round-4 build
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/i_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/i_tests.log
grep -E "^FAILED|^ERROR|Error|assert " gpurun_out/i_tests.log | head -30
for V in "libkvfe_base.so --frames-persist" "libkvfe.so --frames-persist" "libkvfe.so " "libkvfe_base.so --frames-persist" "libkvfe.so --frames-persist" "libkvfe.so "; do
set -- $V
KVFE_LIB=$L/$1 timeout 300 python bench.py --legs nominal,single_stream $2 --steps 52 --warmup 10 --repeats 3 --stage-event-stride 2 > gpurun_out/i_line.json 2> gpurun_out/i_err.log
python - "$1 $2" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], 'value', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (kk[:10], vv) for kk, vv in st.items()))
for k in ('nominal','single_stream'):
    v=d.get(k,{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
    print('   ', k, v.get('value'), v.get('repeats',{}).get('values'))
PY
done
