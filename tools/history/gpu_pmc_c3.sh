# SQ / TCC counters of the c3 main leg alone (kernels serialised by the profiler), plus kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
Q="--steps 10 --warmup 3 --repeats 1 --legs none"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_c3_kf_$C -o p -- python $R/bench.py $Q > $R/gpurun_out/pmc.log 2>&1; echo "pmc c3 $C rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/prof_sq -o s -- python $R/bench.py $Q > $R/gpurun_out/prof_sq.log 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_sq2 -o s2 -- python $R/bench.py $Q > $R/gpurun_out/prof_sq2.log 2>&1; echo "sq2 rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/prof_tcc -o t -- python $R/bench.py $Q > $R/gpurun_out/prof_tcc.log 2>&1; echo "tcc rc=$?"
cd $R
python tools/rocpd_stats.py gpurun_out/prof_kt/kt_results.db | head -30
python tools/rocpd_pmc.py gpurun_out/pmc_c3_kf_FETCH_SIZE/p_results.db gpurun_out/pmc_c3_kf_WRITE_SIZE/p_results.db | head -14
python tools/rocpd_pmc.py gpurun_out/prof_sq/s_results.db | head -14
python tools/rocpd_pmc.py gpurun_out/prof_sq2/s2_results.db | head -14
python tools/rocpd_pmc.py gpurun_out/prof_tcc/t_results.db | head -14
