# round 6: is the headline's 6-8 % deficit against the frames_persist leg the option or the leg's place in the process?
mkdir -p gpurun_out; export TMPDIR=/tmp
for A in "--frames-persist --legs none" "--legs persist" "--legs persist,outputs"; do
timeout 300 python bench.py $A --no-cpu-baseline --repeats 3 > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
fp=d.get("frames_persist") or {}
oi=d.get("outputs_inclusive") or {}
print("$A | value", d.get("value"), d["repeats"]["values"], "enqueue", d.get("host_enqueue_ms_per_step"), "| frames_persist leg", fp.get("repeats",{}).get("values"), fp.get("host_enqueue_ms_per_step"), "| outputs", oi.get("value"), oi.get("no_readback_value"))
PY
done
