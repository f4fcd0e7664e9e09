# round 5, call z: the one-corner cornerSubPix kernel in the headline step is latency bound (40 dependent iterations); an
# iteration takes ~2 x as long inside the step as alone.  Is it the placement (up to 7 blocks of a few hundred on one unit)?
# Extra LDS per block = fewer blocks per unit: 0 (7 per unit), 16 KB (4), 30 KB (3), 42 KB (2).
mkdir -p gpurun_out; export TMPDIR=/tmp
for PAD in 0 30000 16000 42000 0 30000; do
KVFE_SUBPIX_LDS_PAD=$PAD KVFE_SUBPIX_STATS=1 timeout 300 python bench.py --legs none --steps 24 --warmup 30 --repeats 3 --stage-event-stride 1 --no-cpu-baseline > gpurun_out/z_line.json 2> gpurun_out/z_err.log
python - "$PAD" <<'PY'
import json,sys
d=json.load(open('bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('pad', sys.argv[1], 'value', d['value'], d['repeats']['values'], 'subpix %.3f lk %.3f' % (st.get('subpix_append',-1), st.get('lk_track',-1)))
PY
grep KVFE_SUBPIX_STATS gpurun_out/z_err.log | tail -2
done
