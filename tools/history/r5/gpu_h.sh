# round 5, call h: grouped cornerSubPix with as many blocks per stream as the device holds (slots refill); staged-step
# input copy variants; min-eigenvalue launch back in round 4's shape
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/h_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/h_tests.log
grep -E "^FAILED|^ERROR|Error|assert " gpurun_out/h_tests.log | head -30
for V in "libkvfe_base.so --frames-persist" "libkvfe.so --frames-persist" "libkvfe.so " "libkvfe_base.so --frames-persist" "libkvfe.so --frames-persist"; do
set -- $V
KVFE_LIB=$L/$1 timeout 300 python bench.py --legs kf_realistic,c5 $2 --steps 52 --warmup 10 --repeats 3 --stage-event-stride 2 > gpurun_out/h_line.json 2> gpurun_out/h_err.log
python - "$1 $2" <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], 'value', d['value'], d['repeats']['values'], ' '.join('%s %.3f' % (kk[:10], vv) for kk, vv in st.items()))
for k in ('kf_realistic','c5'):
    v=d.get(k,{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
    print('   ', k, v.get('value'), v.get('repeats',{}).get('values'), ' '.join('%s %.3f' % (kk[:10], vv) for kk, vv in st.items()))
PY
done
echo "--- staged (pcie) leg"
python - <<'PY'
import os, subprocess, json
for tag, env in (("ring on copy stream", {}), ("ring on own stream", {"KVFE_X_IN_STREAM": "1"}), ("no ring", {"KVFE_X_IN_RING": "0"}), ("ring on copy stream", {}), ("ring on own stream", {"KVFE_X_IN_STREAM": "1"})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run(["python", "bench.py", "--legs", "pcie", "--steps", "26", "--warmup", "6", "--repeats", "1", "--no-stage-events"], env=e, capture_output=True, text=True, timeout=600)
    try:
        d = json.load(open("bench_detail.json"))
        p = d.get("pcie_inclusive", {})
        print(tag, p.get("value"), p.get("ms_per_step"), "cyclic3", p.get("cyclic3_value"), "pageable", p.get("pageable_value"))
    except Exception as ex:
        print(tag, "failed", ex, r.stderr[-400:])
PY
