# round 6: what the stage events cost the headline (stride 0 = none, 8 = default, 1 = every step), output transfer on / off
mkdir -p gpurun_out; export TMPDIR=/tmp
for V in ${VS:-0:0 0:8 0:1 4096:0 4096:8 0:0 0:8}; do
X=${V%%:*}; S=${V##*:}
if [ "$X" = 0 ]; then unset KVFE_OUT_TRANSFER_BYTES; else export KVFE_OUT_TRANSFER_BYTES=$X; fi
if [ "$S" = 0 ]; then A="--no-stage-events"; else A="--stage-event-stride $S"; fi
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 3 $A > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
print("OUT_TRANSFER_BYTES=$X stride=$S value", d.get("value"), d["repeats"]["values"], "ms/step", d.get("ms_per_step"), "enqueue", d.get("host_enqueue_ms_per_step"))
PY
done
