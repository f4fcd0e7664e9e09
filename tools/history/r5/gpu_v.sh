# round 5, call v: the upload stream in the other priority pool of hardware queues (default GPU_MAX_HW_QUEUES)
mkdir -p gpurun_out; export TMPDIR=/tmp
for V in 0 1 0 1; do
  KVFE_X_COPY_PRIO=$V timeout 300 python bench.py --legs pcie --steps 26 --warmup 6 --repeats 1 --no-stage-events --no-cpu-baseline > gpurun_out/v_line.json 2> gpurun_out/v_err.log
  python - "$V" <<'PY'
import json,sys
d=json.load(open('bench_detail.json')); p=d.get('pcie_inclusive',{})
print('upload stream in the other pool=%s' % sys.argv[1], 'value', d.get('value'), 'pcie', p.get('value'), p.get('ms_per_step'), 'cyclic3', p.get('cyclic3_value'), 'pageable', p.get('pageable_value'), 'enqueue', p.get('host_enqueue_ms_per_step'))
PY
done
