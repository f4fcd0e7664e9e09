# round 6: kernel timeline of a few steady steps of the headline (which stream waits for what)
# usage: bash tools/r6/gpu_timeline.sh <tag> [bench args]
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
tag=${1:-tl}; shift
cd /tmp
env $ENVS timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$tag -o s -- python $R/bench.py --steps 12 --warmup 4 --repeats 1 --legs none --no-stage-events --no-cpu-baseline "$@" > $R/gpurun_out/tl_$tag.log 2>&1; echo "rc=$?"
db=$(find $R/gpurun_out/tl_$tag -name "*.db" | head -1)
python - $db > $R/gpurun_out/tl_$tag.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
def short(n): return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("kvfe::", "")
# the window: from the 9th lk4_kernel launch to the 12th
lk = [i for i, r in enumerate(rows) if short(r[0]).startswith("lk4_kernel")]
a, b = lk[8], lk[11]
t0 = rows[a][1]
for name, s, e, q, st in rows[a - 2:b + 1]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  (+{(e - s) / 1e3:7.1f})  q{q} s{st}  {short(name)}")
PY
cat $R/gpurun_out/tl_$tag.txt | head -${N:-120}
rm -rf $R/gpurun_out/tl_$tag
