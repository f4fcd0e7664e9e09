# round 6: dispatch order of the LK blocks on the other legs (real frames, reference cadence, single stream, 1280x720)
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
timeout 600 python bench.py --legs kf_realistic,nominal,c5,single_stream --no-cpu-baseline --repeats 1 --no-dense > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
print("$1 value", d.get("value"), {k: (d.get(k) or {}).get("value") for k in ("kf_realistic","nominal","c5","single_stream","single_stream_spinonce")})
PY
}
for V in ${VS:-3 0 3 0}; do
export KVFE_LK4_ORDER=$V
run "KVFE_LK4_ORDER=$V"
done
