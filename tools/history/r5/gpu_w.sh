# round 5, call w: does the number of hardware queues matter for the device-fed step (three streams per context)?
mkdir -p gpurun_out; export TMPDIR=/tmp
for Q in "" 8 "" 8; do
  if [ -n "$Q" ]; then export GPU_MAX_HW_QUEUES=$Q; else unset GPU_MAX_HW_QUEUES; fi
  timeout 300 python bench.py --legs kf_realistic,c5 --no-cpu-baseline > gpurun_out/w_line.json 2> gpurun_out/w_err.log
  python - "$Q" <<'PY'
import json,sys
d=json.load(open('bench_detail.json'))
print('GPU_MAX_HW_QUEUES=%s' % (sys.argv[1] or 'default'), 'value', d.get('value'), d.get('repeats',{}).get('values'), 'kf_realistic', d.get('kf_realistic',{}).get('value'), 'c5', d.get('c5',{}).get('value'))
PY
done
