# round 5, call x: cornerSubPix refills off the block's critical path (corner on deck, one-round-trip stage, bearing vectors in
# the commit kernel), track_finalize as one 1024-thread pass, min-eigenvalue mask-phase trims -- parity suite, then A/B
# against the build of 51dd374 (libkvfe_base.so) in the same call
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/x_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/x_tests.log
grep -E "^FAILED|^ERROR" gpurun_out/x_tests.log | head -30
for V in libkvfe_base.so libkvfe.so libkvfe_base.so libkvfe.so; do
KVFE_LIB=$L/$V timeout 300 python bench.py --legs kf_realistic --steps 52 --warmup 10 --repeats 3 --stage-event-stride 2 --no-cpu-baseline > gpurun_out/x_line.json 2> gpurun_out/x_err.log
python - "$V" <<'PY'
import json,sys
d=json.load(open('bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
keys=('lk_track','track_finalize','ransac_mono','mineig_localmax','gftt_select','subpix_append')
print(sys.argv[1], 'value', d['value'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:10], st.get(k, -1)) for k in keys))
v=d.get('kf_realistic',{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
print('    kf_realistic', v.get('value'), v.get('repeats',{}).get('values'), ' '.join('%s %.3f' % (k[:10], st.get(k, -1)) for k in keys))
PY
done
echo "--- new corners per stream, real frames; phase cycles of the grouped kernel"
KVFE_SUBPIX_STATS=1 timeout 200 python tools/r5/spg_probe.py 2>&1 | tail -16
KVFE_LIB=$L/libkvfe_base.so KVFE_SUBPIX_STATS=1 PROBE_N=6 timeout 200 python tools/r5/spg_probe.py 2>&1 | tail -3
