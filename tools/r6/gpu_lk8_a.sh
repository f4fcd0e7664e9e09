# round 6, call a: first run of the eight-points-per-wave LK kernel -- parity on the LK / sequence / headline tests,
# then the LK stage of the headline leg against the one-point kernel (KVFE_LK_IMPL=1) in the same call
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8
for I in 0 1 0 1; do
KVFE_LK_IMPL=$I timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 > gpurun_out/lk8_$I.json 2> gpurun_out/lk8_$I.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("KVFE_LK_IMPL=$I value", d.get("value"), "ms/step", d.get("ms_per_step"), "lk_track", st.get("lk_track"))
PY
done
