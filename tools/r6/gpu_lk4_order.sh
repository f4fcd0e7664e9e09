# round 6: dispatch order of the LK blocks (where do the waves with 30-iteration points start?)
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("$1 value", d.get("value"), "ms/step", d.get("ms_per_step"), "lk_track", st.get("lk_track"))
PY
}
for V in ${VS:-0 1 2 3 0 1 2 3}; do
export KVFE_LK4_ORDER=$V
run "KVFE_LK4_ORDER=$V"
done
