# round 6: iteration cap of the eight-point LK waves (KVFE_LK8_CAP; points past it go to one-point waves) -- parity at the
# default, then the LK stage per cap
mkdir -p gpurun_out; export TMPDIR=/tmp
KVFE_LK_IMPL=0 timeout 900 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
run() {
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 > gpurun_out/lk8_cap.json 2> gpurun_out/lk8_cap.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("$1 value", d.get("value"), "ms/step", d.get("ms_per_step"), "lk_track", st.get("lk_track"))
PY
}
for CAP in ${CAPS:-3 4 5 6 8 12 30}; do
export KVFE_LK_IMPL=0 KVFE_LK8_CAP=$CAP
run "cap $CAP"
done
export KVFE_LK_IMPL=1
run "one-point kernel"
