# round 6: waves per corner of the one-corner cornerSubPix kernel at 64 streams (KVFE_SUBPIX_NW = 2 | 4: the switch was removed with the
# variant's loss -- 0.45 against 0.40 ms; a record, not a runnable A/B)
mkdir -p gpurun_out; export TMPDIR=/tmp
for V in ${VS:-2 4 2 4}; do
export KVFE_SUBPIX_NW=$V
timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("NW=$V value", d.get("value"), "ms/step", d.get("ms_per_step"), "subpix_append", st.get("subpix_append"), "copy", d.get("hbm_copy_GBps"))
PY
done
