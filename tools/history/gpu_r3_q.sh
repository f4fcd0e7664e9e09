# round 3, call q: corner slots per stream of the cornerSubPix launch (KVFE_SUBPIX_SLOTS; default = the bound, 793)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
run() {
env $1 timeout 300 python bench.py --legs $2 --steps 40 --warmup 8 --repeats 2 2> gpurun_out/q_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('nominal','c5','kf_realistic','klt_max_level_4','single_stream') if k in d], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items() if k[:3] in ('sub','rec','ste')))"
}
run KVFE_X=0 kf_realistic
run KVFE_SUBPIX_SLOTS=128 kf_realistic
run KVFE_SUBPIX_SLOTS=64 kf_realistic
run KVFE_SUBPIX_SLOTS=32 kf_realistic
run KVFE_X=0 none
run KVFE_SUBPIX_SLOTS=64 none
