# round 6: does splitting the batch into stream groups (each on its own HIP stream) hide the tail of the eight-point LK
# launch behind the other groups' kernels?
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 --no-stage-events $2 > gpurun_out/lk8_g.json 2> gpurun_out/lk8_g.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
print("$1 value", d.get("value"), "ms/step", d.get("ms_per_step"))
PY
}
export KVFE_LK8_CAP=30
for G in 1 2 4; do
export KVFE_LK_IMPL=0
run "lk8 groups $G" "--groups $G"
export KVFE_LK_IMPL=1
run "one-point groups $G" "--groups $G"
done
