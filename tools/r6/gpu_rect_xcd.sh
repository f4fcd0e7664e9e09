# round 6: block order of rectify_tile_kernel (KVFE_RECT_XCD: 0 = 3-D grid, 1 = XCD-banded tile rows, 2 = an XCD per stream group)
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$PYT" ]; then KVFE_RECT_XCD=2 timeout 900 python -m pytest tests/test_gpu_bench_configs.py -x -q -k "c3" 2>&1 | tail -2; fi
for V in ${VS:-0 1 2 0 1 2}; do
export KVFE_RECT_XCD=$V
timeout 300 python bench.py --legs alone --no-cpu-baseline --repeats 3 > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
al={k["kernel"]: k["avg_launch_ms"] for k in (d.get("dense_kernels_alone") or {}).get("kernels",[])}
print("RECT_XCD=$V value", d.get("value"), d["repeats"]["values"], "copy", d.get("hbm_copy_GBps"), "rectify in-step", st.get("rectify"), "alone", al.get("rectify"))
PY
done
