# round 6: the four-points-per-wave LK kernel -- parity, then the LK stage against the one-point kernel in the same call
mkdir -p gpurun_out; export TMPDIR=/tmp
KVFE_LK_IMPL=0 timeout 900 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6
run() {
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 > gpurun_out/lk4.json 2> gpurun_out/lk4.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("$1 value", d.get("value"), "ms/step", d.get("ms_per_step"), "lk_track", st.get("lk_track"))
PY
}
for I in ${IMPLS:-0 1 0 1}; do
export KVFE_LK_IMPL=$I
run "KVFE_LK_IMPL=$I"
done
