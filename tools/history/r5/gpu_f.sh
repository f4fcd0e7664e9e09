# round 5, call f: split tracking with the refinement on a high-priority stream / raised wave priority; a step's inputs by an
# early copy into a device ring (no kernel reads host memory); staged steps forked; default configuration arranged like
# device_frames_persist
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/f_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/f_tests.log
grep -E "^FAILED|^ERROR|Error|assert " gpurun_out/f_tests.log | head -30
run() {  # label, extra env, extra flags
env $2 timeout 300 python bench.py --legs none $3 --steps 52 --warmup 10 --repeats 3 --stage-event-stride 2 > gpurun_out/f_line.json 2> gpurun_out/f_err.log
python - "$1" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:10], v) for k, v in st.items()))
PY
}
run base-r4 "KVFE_LIB=$L/libkvfe_base.so" "--frames-persist"
run nosplit "KVFE_X_SPLIT=0" "--frames-persist"
run nosplit-noring "KVFE_X_SPLIT=0 KVFE_X_IN_RING=0" "--frames-persist"
run split-prio11 "" "--frames-persist"
run split-prio10 "KVFE_X_SUBPIX_WPRIO=0" "--frames-persist"
run split-prio01 "KVFE_X_SUB_PRIO=0" "--frames-persist"
run split-prio00 "KVFE_X_SUB_PRIO=0 KVFE_X_SUBPIX_WPRIO=0" "--frames-persist"
run split-prio11 "" "--frames-persist"
run default-config-split "" ""
run default-config-nosplit "KVFE_X_SPLIT=0" ""
echo "--- staged (pcie) leg"
python - <<'PY'
import os, subprocess, json
for tag, env in (("fork+ring", {}), ("serial+ring", {"KVFE_X_STAGED_SERIAL": "1"}), ("serial noring", {"KVFE_X_STAGED_SERIAL": "1", "KVFE_X_IN_RING": "0"}),
                 ("fork+ring nosplit", {"KVFE_X_SPLIT": "0"})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run(["python", "bench.py", "--legs", "pcie", "--steps", "26", "--warmup", "6", "--repeats", "1", "--no-stage-events"], env=e, capture_output=True, text=True, timeout=600)
    try:
        d = json.load(open("bench_detail.json"))
        p = d.get("pcie_inclusive", {})
        print(tag, p.get("value"), p.get("ms_per_step"), "cyclic3", p.get("cyclic3_value"), "pageable", p.get("pageable_value"), "link", p.get("link"))
    except Exception as ex:
        print(tag, "failed", ex, r.stderr[-400:])
PY
echo "--- kf_realistic / c5 / nominal with split"
timeout 300 python bench.py --legs kf_realistic,c5,nominal,single_stream --frames-persist --steps 52 --warmup 10 --repeats 2 --stage-event-stride 2 > gpurun_out/f_line.json 2> gpurun_out/f_err.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for k in ('kf_realistic','c5','single_stream','nominal'):
    v=d.get(k,{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
    print(k, v.get('value'), ' '.join('%s %.3f' % (kk[:10], vv) for kk, vv in st.items() if kk in ('lk_track','lk_track_new','subpix_append','mineig_localmax')))
PY
