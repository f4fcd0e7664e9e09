# round 6: phase stamps / wave lifetimes of the eight-points-per-wave LK kernel (libkvfe_lk8prof.so = -DKVFE_LK8_PROF build),
# then the LK stage of the product build against the one-point kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
if [ -f $L/libkvfe_lk8prof.so ]; then
KVFE_LIB=$L/libkvfe_lk8prof.so KVFE_LK_IMPL=0 timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 --steps 20 > gpurun_out/lk8_prof.json 2> gpurun_out/lk8_prof.err
grep KVFE_LK8_PROF gpurun_out/lk8_prof.err
fi
for I in ${IMPLS:-0 1 0 1}; do
KVFE_LK_IMPL=$I timeout 300 python bench.py --legs none --no-cpu-baseline --repeats 1 > gpurun_out/lk8_$I.json 2> gpurun_out/lk8_$I.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
st=d.get("stage_ms_per_step_summed_over_groups",{})
print("KVFE_LK_IMPL=$I value", d.get("value"), "ms/step", d.get("ms_per_step"), "lk_track", st.get("lk_track"))
PY
done
