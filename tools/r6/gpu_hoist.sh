# round 6: the next frame's pyramid on its own stream behind the previous tracking launch (KVFE_PYR_HOIST=0: on the main
# stream between the refinement and the tracking launch, as until round 5)
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$PYT" ]; then timeout 900 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_replay_r3.py tests/test_gpu_pyramid_r3.py -x -q 2>&1 | tail -2; fi
for V in ${VS:-0 1 0 1}; do
export KVFE_PYR_HOIST=$V
timeout 300 python bench.py --legs kf_realistic,persist --no-cpu-baseline --repeats 3 $BARGS > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("PYR_HOIST=$V value", d.get("value"), d["repeats"]["values"], "ms/step", d.get("ms_per_step"), {k: st.get(k) for k in ("pyramid","lk_track","track_finalize","subpix_append")}, "persist", (d.get("frames_persist") or {}).get("repeats",{}).get("values"), "real", (d.get("kf_realistic") or {}).get("value"))
PY
done
