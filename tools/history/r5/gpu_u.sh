# round 5, call u: does the staged-input leg overlap upload and step when the runtime has more hardware queues?
# (five HIP streams per context on GPU_MAX_HW_QUEUES = 4 by default: two of them share a queue)
mkdir -p gpurun_out; export TMPDIR=/tmp
for Q in "" 8 2; do
  if [ -n "$Q" ]; then export GPU_MAX_HW_QUEUES=$Q; else unset GPU_MAX_HW_QUEUES; fi
  timeout 300 python bench.py --legs pcie --steps 26 --warmup 6 --repeats 1 --no-stage-events --no-cpu-baseline > gpurun_out/u_line.json 2> gpurun_out/u_err.log
  python - "$Q" <<'PY'
import json,sys
d=json.load(open('bench_detail.json')); p=d.get('pcie_inclusive',{})
print('GPU_MAX_HW_QUEUES=%s' % (sys.argv[1] or 'default'), 'value', d.get('value'), 'pcie', p.get('value'), p.get('ms_per_step'), 'cyclic3', p.get('cyclic3_value'), 'pageable', p.get('pageable_value'), 'enqueue', p.get('host_enqueue_ms_per_step'))
PY
done
