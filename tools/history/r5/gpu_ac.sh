# round 5, call ac: the chain wave of the grouped cornerSubPix kernel waits for LDS (gpu_ab.sh: the chunk loop takes 10.0 k
# cycles per iteration, 4.9 k when the chain adds nothing) -- reads 16 terms ahead (libkvfe.so) against 24 / 28 (the most
# lgkmcnt can track; 15 / 23 spilled registers)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
for V in libkvfe.so libkvfe_pf28.so libkvfe_pf24.so libkvfe.so libkvfe_pf28.so; do
KVFE_LIB=$L/$V KVFE_SUBPIX_STATS=1 timeout 300 python bench.py --legs kf_realistic --steps 26 --warmup 10 --repeats 2 --stage-event-stride 2 --no-cpu-baseline > gpurun_out/ac_line.json 2> gpurun_out/ac_err.log
python - "$V" <<'PY'
import json,sys
d=json.load(open('bench_detail.json'))
v=d.get('kf_realistic',{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], 'value', d['value'], '  kf_realistic', v.get('value'), v.get('repeats',{}).get('values'), 'subpix %.3f' % st.get('subpix_append', -1))
PY
grep "KVFE_SUBPIX_STATS (group" gpurun_out/ac_err.log | tail -1 | cut -c30-
done
