# round 3, last call: the final build -- full GPU suite, smoke(), one bench line (main leg + kf_realistic + single stream)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --legs kf_realistic,single_stream,nominal > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/final_bench.json'))
print("value", d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('nominal','single_stream','kf_realistic') if k in d])
print("stages", d.get('stage_ms_per_step_summed_over_groups'))
print("roofline", d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
