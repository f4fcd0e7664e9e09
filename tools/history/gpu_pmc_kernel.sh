# SQ counters of the c3 main leg for each variant in $VARIANTS ('|' separated, ';' separated VAR=value inside); prints the
# rows of the kernels matching $KERNELS (grep -E pattern)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
Q="--steps 8 --warmup 3 --repeats 1 --legs none"
IFS='|' read -ra BV <<< "${VARIANTS:-}"
for v in "${BV[@]}"; do
  ( IFS=';'; for kv in $v; do [ -n "$kv" ] && export "$kv"; done; IFS=$' \t\n'
    cd /tmp
    rm -rf $R/gpurun_out/pk_sq $R/gpurun_out/pk_sq2
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/pk_sq -o s -- python $R/bench.py $Q ${BENCH_ARGS:-} > $R/gpurun_out/pk_sq.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS -d $R/gpurun_out/pk_sq2 -o s2 -- python $R/bench.py $Q ${BENCH_ARGS:-} > $R/gpurun_out/pk_sq2.log 2>&1
    cd $R
    echo "=== [$v]"
    python tools/rocpd_pmc.py gpurun_out/pk_sq/s_results.db | grep -E "kernel \||${KERNELS:-.}"
    python tools/rocpd_pmc.py gpurun_out/pk_sq2/s2_results.db | grep -E "kernel \||${KERNELS:-.}" )
done
