# round 6: lk4 -- iteration cap sweep (points past the cap go to one-point waves), then the whole GPU suite on the default
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 > gpurun_out/lk4.json 2> gpurun_out/lk4.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("$1 value", d.get("value"), "ms/step", d.get("ms_per_step"), "lk_track", st.get("lk_track"))
PY
}
for CAP in ${CAPS:-30 4 6 8 12 30}; do
export KVFE_LK8_CAP=$CAP
run "cap $CAP"
done
unset KVFE_LK8_CAP
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
