# one GPU call of round 2: parity tests, the full bench line, then (PROFILE=1) the rocprofv3 passes:
# kernel trace of the default bench and per-leg FETCH_SIZE / WRITE_SIZE passes (each leg alone).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -15 gpurun_out/gpu_tests.log
timeout 900 python bench.py $BENCH_ARGS > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print("value", d['value'], d['ms_per_step'], d.get('repeats'))
print("stages", d.get('stage_ms_per_step_summed_over_groups'))
for k in ('nominal','single_stream','c5','dense_stereo','dense_stereo_c5','pcie_inclusive','cpu_baseline'):
    v=d.get(k)
    if v: print(k, v.get('value'), v.get('ms_per_step', v.get('ms_per_pair')), v.get('check'), v.get('stage_ms_per_step_summed_over_groups'))
print("roofline", d.get('roofline'))
PY
if [ -n "$PROFILE" ]; then
cd /tmp
Q="--steps 10 --warmup 3 --repeats 1"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_c5 -o kt -- python $R/bench.py --config c5 --steps 10 --warmup 3 --repeats 1 --legs none > $R/gpurun_out/prof_kt_c5.log 2>&1; echo "kt c5 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_c2 -o kt -- python $R/bench.py --config c2 --steps 200 --warmup 20 --repeats 1 --legs none --no-stage-events > $R/gpurun_out/prof_kt_c2.log 2>&1; echo "kt c2 rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_c3_kf_$C -o p -- python $R/bench.py $Q --legs none > $R/gpurun_out/pmc.log 2>&1; echo "pmc c3 $C rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_c5_kf_$C -o p -- python $R/bench.py --config c5 $Q --legs none > $R/gpurun_out/pmc.log 2>&1; echo "pmc c5 $C rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_dense_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --legs dense > $R/gpurun_out/pmc.log 2>&1; echo "pmc dense $C rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_dense_c5_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --legs dense_c5 > $R/gpurun_out/pmc.log 2>&1; echo "pmc dense_c5 $C rc=$?"
done
if [ -n "$SQ" ]; then
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof_sq -o s -- python $R/bench.py $Q --legs none > $R/gpurun_out/prof_sq.log 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_sq2 -o s2 -- python $R/bench.py $Q --legs none > $R/gpurun_out/prof_sq2.log 2>&1; echo "sq2 rc=$?"
fi
cd $R
# keep only the databases (the traces' other outputs are large)
find gpurun_out -name "*.db" | head -40
du -sh gpurun_out
fi
