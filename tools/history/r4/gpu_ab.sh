# round 4, call ab: tracking launch at 7 waves per SIMD (__launch_bounds__(64, 7): 72 VGPRs, 8 spilled) against 6 (80, none)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
KVFE_LIB=$L/libkvfe_w7.so timeout 600 python -m pytest tests/test_gpu_bench_configs.py -m gpu -q -x -k "c3_headline or c2_single or klt" 2>&1 | tail -2
for V in libkvfe.so libkvfe_w7.so libkvfe.so libkvfe_w7.so; do
KVFE_LIB=$L/$V timeout 300 python bench.py --legs klt4 --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/ab_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$V]', d['value'], d['ms_per_step'], d['repeats']['values'], 'lk %.3f' % st['lk_track'], 'klt4', d['klt_max_level_4']['value'], 'lk %.3f' % d['klt_max_level_4']['stage_ms_per_step_summed_over_groups']['lk_track'])
"
done
