# clean single-stream (BASELINE configs[1]) profile: kernel trace of `bench.py --config c2` alone + its stage timers
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --config c2 --legs none --steps 200 --warmup 20 --repeats 3 --stage-event-stride 1 > gpurun_out/c2_stage.json 2> gpurun_out/c2_stage.err
python bench.py --config c2 --legs none --steps 200 --warmup 20 --repeats 3 --no-stage-events > gpurun_out/c2_plain.json 2> gpurun_out/c2_plain.err
python - <<'PY'
import json
for f in ('gpurun_out/c2_stage.json', 'gpurun_out/c2_plain.json'):
    try:
        d = json.load(open(f))
        print(f, d['value'], d['ms_per_step'], 'enq', d.get('host_enqueue_ms_per_step'), d.get('stage_ms_per_step_summed_over_groups'))
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.json', '.err')).read()[-800:])
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c2_kt -o c2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config c2 --legs none --steps 200 --warmup 20 --repeats 1 --no-stage-events > $GRAFT_REPO_ROOT/gpurun_out/c2_kt.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/c2_kt/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steady state: last 60 steps worth of dispatches; print one step's timeline
names = [r['Kernel_Name'].replace('void ', '').replace('kvfe::', '').split('(')[0][:40] for r in rows]
# find indices of pyrdown pairs as step starts
starts = [i for i, n in enumerate(names) if n.startswith('pyrdown') and (i == 0 or not names[i-1].startswith('pyrdown'))]
i0, i1 = starts[-20], starts[-19]
t0 = int(rows[i0]['Start_Timestamp'])
for r, n in zip(rows[i0:i1], names[i0:i1]):
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%-42s start %8.1f us  dur %7.1f us  grid %s wg %s" % (n, s / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?')))
print('step period us', (int(rows[starts[-19]]['Start_Timestamp']) - t0) / 1e3)
agg = collections.defaultdict(list)
for r, n in zip(rows[starts[-150]:], names[starts[-150]:]):
    agg[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-42s n %5d avg %7.2f us  per-step %7.2f" % (n, len(v), sum(v) / len(v), sum(v) / 150)); tot += sum(v) / 150
print('kernel sum per step us', tot)
PY
