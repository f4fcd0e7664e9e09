# round 6: pyr2_kernel strip height (KVFE_PYR2_T2 = second-level rows per strip: 4 = 1920 waves at 64 x 752x480,
# 2 = 3840, 1 = 7680) -- pyramid stage in the step and alone, beside the copy probe of the same process
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
timeout 300 python bench.py --legs alone --no-cpu-baseline --repeats 1 > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
al={k["kernel"]: k["avg_launch_ms"] for k in (d.get("dense_kernels_alone") or {}).get("kernels",[])}
print("$1 value", d.get("value"), "copy", d.get("hbm_copy_GBps"), "in-step", {k: st.get(k) for k in ("pyramid","rectify","mineig_localmax")}, "alone", al)
PY
}
for V in ${VS:-4 2 1 4 2 1}; do
export KVFE_PYR2_T2=$V
[ -n "$PYT" ] && timeout 300 python -m pytest tests/test_gpu_pyramid_r3.py -x -q 2>&1 | tail -2
run "KVFE_PYR2_T2=$V"
done
