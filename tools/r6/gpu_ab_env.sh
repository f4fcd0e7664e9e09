# round 6: generic A/B of one environment switch on the headline (+ real frames): VAR=name VS="0 1 0 1" [LEGS=kf_realistic]
mkdir -p gpurun_out; export TMPDIR=/tmp
for V in ${VS:-0 1 0 1}; do
export $VAR=$V
timeout 300 python bench.py --legs ${LEGS:-kf_realistic} --no-cpu-baseline --repeats 3 $BARGS > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
kr=d.get("kf_realistic") or {}
print("$VAR=$V value", d.get("value"), d["repeats"]["values"], "ms/step", d.get("ms_per_step"), {k: st.get(k) for k in "${STAGES:-lk_track subpix_append stereo_match mineig_localmax ransac_mono}".split()}, "| real", kr.get("value"), {k: (kr.get("stage_ms_per_step_summed_over_groups") or {}).get(k) for k in "${STAGES:-lk_track subpix_append}".split()})
PY
done
