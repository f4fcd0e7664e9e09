# round 5, call n: SQ counters of the two-pass aggregation kernel (8 pairs of 752x480), three --pmc passes
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/n_sq$i -o s -- python $R/tools/r5/dense_probe.py 8 > $R/gpurun_out/n_sq$i.log 2>&1; echo "pass $i rc=$?"
  db=$(find $R/gpurun_out/n_sq$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db | grep -E "^\| kernel|^\|---|dense_aggregate_pass|dense_cost_fused|dense_select" 
done
