# round 6: which runtime call holds the host for 6.5 ms inside the output hipMemcpyAsync?  (HSA + HIP API trace, calls > 2 ms)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
cd /tmp
KVFE_DMA_WARMUP=0 KVFE_HOST_PROF=1 timeout 300 rocprofv3 --hsa-trace --hip-runtime-trace -d $R/gpurun_out/hsa -o h -- python $R/tools/r6/stall_probe.py none > $R/gpurun_out/hsa.log 2>&1; echo rc=$?
grep "per call" $R/gpurun_out/hsa.log | python -c "
import sys
for l in sys.stdin:
    v=[float(x) for x in l.split(':')[1].split()]
    print('stalls', [(i,int(x)) for i,x in enumerate(v) if x>1000], 'of', len(v))"
db=$(find $R/gpurun_out/hsa -name "*.db" | head -1)
python - $db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'region' in t.lower() or 'api' in t.lower()][:20])
for t in ("regions_and_samples", "regions"):
    if t in tabs:
        cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
        print(t, cols)
        try:
            rows = c.execute(f"select name, start, end, (end-start) as d, category from {t} where (end-start) > 2000000 order by start").fetchall()
        except Exception as e:
            print("query failed", e); continue
        t0 = rows[0][1] if rows else 0
        for r in rows[:80]:
            print(f"{(r[1]-t0)/1e6:10.2f} ms  +{r[3]/1e6:8.2f} ms  {r[4]}  {r[0]}")
        break
PY
rm -rf $R/gpurun_out/hsa
