# round 5, call aa: grouped cornerSubPix -- producers read their window pixel's weight, patch offset and coordinates as ONE
# table word pair (no division per chunk, no padding branch), patch entries take their left neighbour's running term through
# a DPP wave shift (two stage bytes per entry instead of four): parity of the configurations that use the kernel, then A/B
# against the build of the final profiles (libkvfe_r5v3.so)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 600 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_pipelined_r3.py -m gpu -q > gpurun_out/aa_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/aa_tests.log
grep -E "^FAILED|^ERROR" gpurun_out/aa_tests.log | head -30
for V in libkvfe_r5v3.so libkvfe.so libkvfe_r5v3.so libkvfe.so; do
KVFE_LIB=$L/$V KVFE_SUBPIX_STATS=1 timeout 300 python bench.py --legs kf_realistic --steps 26 --warmup 10 --repeats 3 --stage-event-stride 2 --no-cpu-baseline > gpurun_out/aa_line.json 2> gpurun_out/aa_err.log
python - "$V" <<'PY'
import json,sys
d=json.load(open('bench_detail.json'))
v=d.get('kf_realistic',{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], 'value', d['value'], '  kf_realistic', v.get('value'), v.get('repeats',{}).get('values'), 'subpix %.3f' % st.get('subpix_append', -1))
PY
grep "KVFE_SUBPIX_STATS (group" gpurun_out/aa_err.log | tail -1
done
