# round 4, call ae (experiment): waves of mineig2_kernel with many needed rows at raised issue priority (s_setprio)
# (result: 64 x 752x480 0.0786 -> 0.0732 ms, slowest wave 175 -> 154 k cycles, step +0.8 %; real frames 0.089 -> 0.091 ms, 1280x720
# 0.112 -> 0.118 ms: mixed, not kept -- the KVFE_ME_PRIO code is not in the tree)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
for V in libkvfe.so libkvfe_DKVFE_ME_PRIO.so libkvfe.so libkvfe_DKVFE_ME_PRIO.so; do
KVFE_LIB=$L/$V timeout 300 python bench.py --legs kf_realistic,c5 --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/ae_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$V]', d['value'], d['ms_per_step'], d['repeats']['values'], 'mineig %.4f' % st['mineig_localmax'])
for k in ('kf_realistic','c5'):
    print('   ', k, d[k]['value'], 'mineig %.4f' % d[k]['stage_ms_per_step_summed_over_groups']['mineig_localmax'])
"
done
KVFE_LIB=$L/libkvfe_DKVFE_ME_PRIODKVFE_ME_PROF.so timeout 300 python bench.py --legs none --steps 20 --warmup 5 --repeats 1 --no-stage-events 2>&1 >/dev/null | grep KVFE_ME_PROF
