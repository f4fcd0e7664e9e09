# one GPU call: parity tests, bench line, kernel trace and PMC passes (all G=default)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/bench.json
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-stream"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- $B > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o f -- $B > $R/gpurun_out/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o w -- $B > $R/gpurun_out/prof_write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof_sq -o s -- $B > $R/gpurun_out/prof_sq.log 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_sq2 -o s2 -- $B > $R/gpurun_out/prof_sq2.log 2>&1; echo "sq2 rc=$?"
cd $R
