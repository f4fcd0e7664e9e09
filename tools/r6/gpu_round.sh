# one GPU call of round 6 on the final build (successor of tools/history/r5/gpu_round.sh): sanity, parity tests, the full bench line, then (PROFILE=1) the rocprofv3
# passes -- kernel traces (c3 / c5 / c2), FETCH_SIZE / WRITE_SIZE per leg (each leg alone), SQ counters of the c3 leg.
# Every leg has its own short timeout (a wedged box must not eat the budget).
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -4 gpurun_out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py $BENCH_ARGS > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$? line bytes $(wc -c < gpurun_out/bench_line.json)"
cat gpurun_out/bench_line.json
cp bench_detail.json gpurun_out/bench.json
python - <<'PY'
import json
json.loads(open('gpurun_out/bench_line.json').read())   # the stdout line parses
d=json.load(open('gpurun_out/bench.json'))
print("value", d['value'], d['ms_per_step'], d.get('repeats', {}).get('values'))
print("stages", d.get('stage_ms_per_step_summed_over_groups'))
for k in ('frames_persist','first_steps','nominal','single_stream','c5','klt_max_level_4','kf_realistic','outputs_inclusive','single_stream_spinonce','dense_stereo','dense_stereo_c5','pcie_inclusive','cpu_baseline'):
    v=d.get(k)
    if v: print(k, v.get('value'), v.get('ms_per_step', v.get('ms_per_pair')), v.get('vs_no_readback'), v.get('check'))
print("roofline", d.get('roofline'))
print("weighted", d.get('roofline_dense_weighted'))
for k in d.get('roofline_kernels', []): print("   ", k['kernel'], k['frac'], k['avg_launch_ms'], k.get('traffic'))
print("alone", d.get('dense_kernels_alone'))
print("pcie", d.get('pcie_inclusive'))
PY
if [ -n "$PROFILE" ]; then
cd /tmp
Q="--steps 10 --warmup 3 --repeats 1"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_c5 -o kt -- python $R/bench.py --config c5 --steps 10 --warmup 3 --repeats 1 --legs none > $R/gpurun_out/prof_kt_c5.log 2>&1; echo "kt c5 rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_e -o kt -- python $R/bench.py --steps 1 --warmup 1 --repeats 1 --legs kf_realistic --no-stage-events > $R/gpurun_out/prof_kt_e.log 2>&1; echo "kt kf_realistic rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_c2 -o kt -- python $R/bench.py --config c2 --steps 200 --warmup 20 --repeats 1 --legs none --no-stage-events > $R/gpurun_out/prof_kt_c2.log 2>&1; echo "kt c2 rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_c3_kf_$C -o p -- python $R/bench.py $Q --legs none --no-stage-events > $R/gpurun_out/pmc.log 2>&1; echo "pmc c3 $C rc=$?"
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_c5_kf_$C -o p -- python $R/bench.py --config c5 $Q --legs none --no-stage-events > $R/gpurun_out/pmc.log 2>&1; echo "pmc c5 $C rc=$?"
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_dense_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --legs dense > $R/gpurun_out/pmc.log 2>&1; echo "pmc dense $C rc=$?"
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_dense_c5_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --legs dense_c5 > $R/gpurun_out/pmc.log 2>&1; echo "pmc dense_c5 $C rc=$?"
done
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof_sq -o s -- python $R/bench.py $Q --legs none --no-stage-events > $R/gpurun_out/prof_sq.log 2>&1; echo "sq rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS -d $R/gpurun_out/prof_sq2 -o s2 -- python $R/bench.py $Q --legs none --no-stage-events > $R/gpurun_out/prof_sq2.log 2>&1; echo "sq2 rc=$?"
cd $R
find gpurun_out -name "*.db" | wc -l
du -sh gpurun_out
fi
