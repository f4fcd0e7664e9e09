# A/B of kernel variants selected by environment variables: runs the GPU parity tests once per variant listed in
# $TEST_VARIANTS and a short bench (main leg only, stage events on every step) per variant in $VARIANTS.
# A variant is a ';'-separated list of VAR=value, e.g. VARIANTS="KVFE_RECT_IMPL=0|KVFE_RECT_IMPL=1;KVFE_RECT_SPB=4"
mkdir -p gpurun_out
export TMPDIR=/tmp
IFS='|' read -ra TV <<< "${TEST_VARIANTS:-}"
for v in "${TV[@]}"; do
  ( IFS=';'; for kv in $v; do [ -n "$kv" ] && export "$kv"; done; IFS=$' \t\n'
    L=gpurun_out/ab_tests_$(echo "$v" | tr ';=/' '___').log
    timeout 1200 python -m pytest tests -m gpu -x -q ${TEST_ARGS:-} > $L 2>&1
    echo "TESTS [$v] rc=$? $(tail -1 $L)" )
done
IFS='|' read -ra BV <<< "${VARIANTS:-}"
for v in "${BV[@]}"; do
  ( IFS=';'; for kv in $v; do [ -n "$kv" ] && export "$kv"; done; IFS=$' \t\n'
    timeout 600 python bench.py --legs ${LEGS:-none} --repeats ${REPEATS:-2} --steps 20 --warmup 5 --stage-event-stride 1 ${BENCH_ARGS:-} > gpurun_out/ab_bench.json 2> gpurun_out/ab_bench.err
    python - "$v" <<'PY'
import json, sys
try:
    d = json.load(open('gpurun_out/ab_bench.json'))
    st = d.get('stage_ms_per_step_summed_over_groups', {})
    print("BENCH [%s] value %.0f ms/step %.4f | " % (sys.argv[1], d['value'], d['ms_per_step']) + " ".join("%s %.3f" % (k[:10], v) for k, v in st.items()))
    for leg in ('c5', 'single_stream', 'nominal'):
        if leg in d:
            print("   ", leg, d[leg]['value'], d[leg].get('stage_ms_per_step_summed_over_groups'))
except Exception as e:
    print("BENCH [%s] failed: %r" % (sys.argv[1], e)); print(open('gpurun_out/ab_bench.err').read()[-1500:])
PY
  )
done
