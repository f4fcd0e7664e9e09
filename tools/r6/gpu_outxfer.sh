# round 6: does the per-step output transfer (5 MB D2H on the output stream) still delay kernels that end beside it?
# KVFE_OUT_TRANSFER_BYTES=4096 shrinks it to 4 KB (debugging aid; `value` reads no outputs)
mkdir -p gpurun_out; export TMPDIR=/tmp
for V in ${VS:-0 4096 0 4096}; do
if [ "$V" = 0 ]; then unset KVFE_OUT_TRANSFER_BYTES; else export KVFE_OUT_TRANSFER_BYTES=$V; fi
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 3 $BARGS > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("OUT_TRANSFER_BYTES=$V value", d.get("value"), d["repeats"]["values"], "ms/step", d.get("ms_per_step"), {k: st.get(k) for k in ("lk_track","track_finalize","subpix_append","step_finalize")})
PY
done
