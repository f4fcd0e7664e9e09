# round 4, call m: C(p,d) of the dense SGBM path in one kernel (pixel cost + row sums + column sums fused)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py tests/test_gpu_components_r2.py tests/test_gpu_fuzz_slices.py -m gpu -q -x -k "dense or fuzz" > gpurun_out/m_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/m_tests.log | cut -c1-300
timeout 300 python bench.py --legs dense,dense_c5 --steps 10 --warmup 4 --repeats 1 2> gpurun_out/m_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read())
for k in ('dense_stereo','dense_stereo_c5'):
    print(k, {a: d[k].get(a) for a in ('value','ms_per_pair','ms_per_pair_min')})
"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o m -- python $GRAFT_REPO_ROOT/bench.py --legs dense --steps 4 --warmup 2 --repeats 1 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_m/**/*kernel_stats.csv', recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows:
        if 'dense' in r['Name'] or 'speckle' in r['Name']:
            print('%-60s calls %4s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
