# round 5, call g: grouped cornerSubPix with refill (a finished corner's slot takes the stream's next corner); staged steps
# with their inputs copied behind the frames; split tracking removed
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/g_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/g_tests.log
grep -E "^FAILED|^ERROR|Error|assert " gpurun_out/g_tests.log | head -30
for V in "libkvfe_base.so --frames-persist" "libkvfe.so --frames-persist" "libkvfe.so " "libkvfe_base.so --frames-persist" "libkvfe.so --frames-persist"; do
set -- $V
KVFE_LIB=$L/$1 timeout 300 python bench.py --legs kf_realistic,c5,nominal $2 --steps 52 --warmup 10 --repeats 3 --stage-event-stride 2 > gpurun_out/g_line.json 2> gpurun_out/g_err.log
python - "$1 $2" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], 'value', d['value'], d['repeats']['values'], 'subpix %.3f lk %.3f mineig %.3f' % (st['subpix_append'], st['lk_track'], st['mineig_localmax']))
for k in ('kf_realistic','c5','nominal'):
    v=d.get(k,{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
    print('   ', k, v.get('value'), v.get('repeats',{}).get('values'), ' '.join('%s %.3f' % (kk[:10], vv) for kk, vv in st.items() if kk in ('lk_track','subpix_append','mineig_localmax')))
PY
done
echo "--- staged (pcie) leg"
timeout 300 python bench.py --legs pcie --steps 26 --warmup 6 --repeats 1 --no-stage-events > gpurun_out/g_line.json 2> gpurun_out/g_err.log
python -c "
import json; d=json.load(open('bench_detail.json')); print('pcie', d.get('pcie_inclusive'))"
