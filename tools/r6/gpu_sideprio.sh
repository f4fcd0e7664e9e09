# round 6: side stream (rectify / match / reject chain + the step's tail) in the high-priority queue pool (KVFE_SIDE_PRIO=1)
mkdir -p gpurun_out; export TMPDIR=/tmp
for V in ${VS:-0 1 0 1}; do
export KVFE_SIDE_PRIO=$V
timeout 300 python bench.py --legs kf_realistic --no-cpu-baseline --repeats 3 $BARGS > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("SIDE_PRIO=$V value", d.get("value"), d["repeats"]["values"], "ms/step", d.get("ms_per_step"), {k: st.get(k) for k in ("lk_track","track_finalize","mineig_localmax","subpix_append","step_finalize","stereo_match")}, "real", (d.get("kf_realistic") or {}).get("value"))
PY
done
