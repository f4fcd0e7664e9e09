# round 3, call A: instruction-rate microbenchmark, pyramid parity tests, kernel trace + PMC (traffic, SQ) of the c3 leg
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 300 tools/ubench/valu_rate > gpurun_out/valu_rate.txt 2>&1; echo "valu_rate rc=$?"; cat gpurun_out/valu_rate.txt
timeout 600 python -m pytest tests/test_gpu_pyramid_r3.py -x -q > gpurun_out/pyr_tests.log 2>&1; echo "pyr tests rc=$?"; tail -5 gpurun_out/pyr_tests.log
Q="--steps 10 --warmup 3 --repeats 1 --legs none"
cd /tmp
rm -rf $R/gpurun_out/r3a_*
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3a_kt -o kt -- python $R/bench.py $Q --no-stage-events > $R/gpurun_out/r3a_kt.log 2>&1; echo "kt rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r3a_fetch -o p -- python $R/bench.py $Q --no-stage-events > $R/gpurun_out/r3a_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r3a_write -o p -- python $R/bench.py $Q --no-stage-events > $R/gpurun_out/r3a_write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/r3a_sq -o s -- python $R/bench.py $Q --no-stage-events > $R/gpurun_out/r3a_sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR -d $R/gpurun_out/r3a_sq2 -o s2 -- python $R/bench.py $Q --no-stage-events > $R/gpurun_out/r3a_sq2.log 2>&1; echo "sq2 rc=$?"
cd $R
python tools/rocpd_stats.py $(find gpurun_out/r3a_kt -name "*.db" | head -1) | head -30
python tools/rocpd_pmc.py $(find gpurun_out/r3a_fetch -name "*.db" | head -1) | head -12
python tools/rocpd_pmc.py $(find gpurun_out/r3a_write -name "*.db" | head -1) | head -12
python tools/rocpd_pmc.py $(find gpurun_out/r3a_sq -name "*.db" | head -1) | head -12
python tools/rocpd_pmc.py $(find gpurun_out/r3a_sq2 -name "*.db" | head -1) | head -12
