# round 5, call e: (1) min-eigenvalue launch as two persistent blocks per compute unit pulling items off an LDS counter;
# (2) split tracking: the previous frame's old points tracked while its new corners are still being refined
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/e_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/e_tests.log
grep -E "^FAILED|^ERROR|Error|assert " gpurun_out/e_tests.log | head -30
run() {  # lib, label, extra env
env KVFE_LIB=$L/$1 $3 timeout 300 python bench.py --legs none --frames-persist --steps 52 --warmup 10 --repeats 3 --stage-event-stride 2 > gpurun_out/e_line.json 2> gpurun_out/e_err.log
python - "$1 $2" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:10], v) for k, v in st.items()))
PY
grep KVFE_ME_PROF gpurun_out/e_err.log
}
run libkvfe_base.so base ""
run libkvfe.so nosplit-auto "KVFE_X_SPLIT=0"
for R in 120 80 60; do run libkvfe.so nosplit-rows$R "KVFE_X_SPLIT=0 KVFE_ME_ROWS=$R"; done
run libkvfe.so split-auto ""
run libkvfe_base.so base ""
run libkvfe.so split-auto ""
run libkvfe_meprof.so prof-nosplit "KVFE_X_SPLIT=0"
echo "--- kf_realistic / c5 / single stream / nominal"
for V in "libkvfe_base.so 1" "libkvfe.so 0" "libkvfe.so 1"; do
set -- $V
KVFE_LIB=$L/$1 KVFE_X_SPLIT=$2 timeout 300 python bench.py --legs kf_realistic,c5,single_stream,nominal --frames-persist --steps 52 --warmup 10 --repeats 2 --stage-event-stride 2 > gpurun_out/e_line.json 2> gpurun_out/e_err.log
python - "$1 split=$2" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json'))
for k in ('kf_realistic','c5','single_stream','nominal'):
    v=d.get(k,{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
    print(sys.argv[1], k, v.get('value'), ' '.join('%s %.3f' % (kk[:10], vv) for kk, vv in st.items() if kk in ('lk_track','lk_track_new','subpix_append','mineig_localmax')))
PY
done
echo "--- default configuration (no frames-persist)"
KVFE_LIB=$L/libkvfe.so timeout 300 python bench.py --legs none --steps 52 --warmup 10 --repeats 3 --stage-event-stride 2 > gpurun_out/e_line.json 2> gpurun_out/e_err.log; python -c "
import json; d=json.load(open('gpurun_out/bench_detail.json')); print('default', d['value'], d['repeats']['values'])"
