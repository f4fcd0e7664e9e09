# round 6: stream groups re-measured (round 1: one group fastest at every batch size -- with a 0.5 ms tracking launch that
# filled the chip; the launch is 0.26 ms now and the main-stream chain is a string of latency-bound kernels)
mkdir -p gpurun_out; export TMPDIR=/tmp
for G in ${GS:-1 2 4 1 2 4}; do
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 --groups $G --no-stage-events > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
print("groups $G value", d.get("value"), "ms/step", d.get("ms_per_step"), "enqueue", d.get("host_enqueue_ms_per_step"), "copy", d.get("hbm_copy_GBps"))
PY
done
