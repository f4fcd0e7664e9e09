# round 6: min-eigenvalue launch reading the stream's disc bitmap (KVFE_ME_BITMAP=1) against rasterising per wave (=0), and
# strip heights (KVFE_ME_ROWS) once the mask phase is a round trip
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
timeout 300 python bench.py --legs alone,kf_realistic --no-cpu-baseline --repeats 1 > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
al={k["kernel"]: k["avg_launch_ms"] for k in (d.get("dense_kernels_alone") or {}).get("kernels",[])}
kr=d.get("kf_realistic") or {}
print("$1 value", d.get("value"), "copy", d.get("hbm_copy_GBps"), "mineig in-step", st.get("mineig_localmax"), "alone", al.get("mineig_localmax"),
      "| real frames", kr.get("value"), (kr.get("stage_ms_per_step_summed_over_groups") or {}).get("mineig_localmax"))
PY
}
if [ -n "$PYT" ]; then
for R in $PYT; do
KVFE_ME_ROWS=$R timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_detectors_r5.py tests/test_gpu_replay_r3.py -x -q 2>&1 | tail -2
done
fi
for V in ${VS:-0:0 1:0 1:60 1:40 1:30 1:20 0:0 1:0 1:40}; do
export KVFE_ME_BITMAP=${V%%:*} KVFE_ME_ROWS=${V##*:}
run "BITMAP=$KVFE_ME_BITMAP ROWS=$KVFE_ME_ROWS"
done
