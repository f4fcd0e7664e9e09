# round 6: 6.5 - 10 ms stalls inside the output hipMemcpyAsync of a process's first context -- which runtime setting moves them?
mkdir -p gpurun_out; export TMPDIR=/tmp
for E in "X=1" "ROC_SIGNAL_POOL_SIZE=1024" "HSA_ENABLE_SDMA=0" "DEBUG_CLR_LIMIT_BLIT_WG=16" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HSA_ENABLE_INTERRUPT=0" "KVFE_DMA_WARMUP=0"; do
env $E KVFE_HOST_PROF=1 timeout 300 python bench.py --legs persist --no-cpu-baseline --repeats 3 2>gpurun_out/hp.err >gpurun_out/hp.json
python - "$E" <<'PY'
import json, re, sys
d = json.load(open("gpurun_out/bench_detail.json"))
calls = [l for l in open("gpurun_out/hp.err") if "per call" in l]
st = []
for l in calls:
    v = [float(x) for x in l.split(":")[1].split()]
    st.append([(i, int(x)) for i, x in enumerate(v) if x > 1000])
print(sys.argv[1], "| value", d["value"], d["repeats"]["values"], "enq", d["host_enqueue_ms_per_step"], "| persist", d["frames_persist"]["repeats"]["values"], "| stalls ctx1", st[0] if st else None, "ctx2", st[1] if len(st) > 1 else None)
PY
done
