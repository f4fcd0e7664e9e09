# round 4, call j: cornerSubPix occupancy: libkvfe_w3.so = launch bounds (128, 3), libkvfe_w4.so = (128, 4) with 30 scratch
# spills, libkvfe.so = (128, 3) and (256, 3) for the four-wave variant of the few-stream case
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
run() {
KVFE_LIB=$L/$1 KVFE_SUBPIX_GROUP=$3 timeout 300 python bench.py --legs ${2:-none} --steps 30 --warmup 8 --repeats 2 --stage-event-stride 4 2> gpurun_out/j_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$1 group=$3]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('kf_realistic','c5','single_stream'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
}
run libkvfe_w3.so kf_realistic,single_stream 0
run libkvfe_w4.so kf_realistic,single_stream 0
run libkvfe.so single_stream 0
run libkvfe_w4.so kf_realistic -1
