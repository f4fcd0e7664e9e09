#!/bin/bash
# tools/save_profiles_r2.sh <tag>: summarise the rocprofv3 databases of the last tools/gpu_r2.sh PROFILE=1 run
# (gpurun_out/, scratch) into profiles/<tag>_*.{md,json,log} (tracked).
set -e
T=$1
cp gpurun_out/bench.json profiles/${T}_bench.json
cp gpurun_out/gpu_tests.log profiles/${T}_gpu_tests.log
for L in "" _c5 _c2; do
  if [ -f gpurun_out/prof_kt$L/kt_results.db ]; then
  { echo "# rocprofv3 --kernel-trace --stats, bench.py leg ${L:-_c3} alone (see tools/gpu_r2.sh for the command)"; echo;
    python tools/rocpd_stats.py gpurun_out/prof_kt$L/kt_results.db; } > profiles/${T}_rocprof_kernel_stats${L}.md
  fi
done
ARGS=""
for L in c3_kf c5_kf dense dense_c5; do
  if [ -f gpurun_out/pmc_${L}_FETCH_SIZE/p_results.db ]; then
    ARGS="$ARGS --leg $L gpurun_out/pmc_${L}_FETCH_SIZE/p_results.db gpurun_out/pmc_${L}_WRITE_SIZE/p_results.db"
  fi
done
python tools/rocpd_traffic.py $ARGS > profiles/${T}_pmc_traffic.json
cp profiles/${T}_pmc_traffic.json profiles/pmc_traffic_latest.json
if [ -f gpurun_out/prof_sq/s_results.db ]; then
{ echo "# rocprofv3 --pmc SQ passes, bench.py c3 leg alone (SQ_* cycle counters are quad-cycles summed over all SIMDs/XCDs)"; echo;
  python tools/rocpd_pmc.py gpurun_out/prof_sq/s_results.db; echo;
  python tools/rocpd_pmc.py gpurun_out/prof_sq2/s2_results.db 2>/dev/null || true; } > profiles/${T}_pmc_sq.md
python tools/rocpd_pmc.py --json gpurun_out/prof_sq/s_results.db > profiles/pmc_valu_latest.json
fi
ls -la profiles/${T}_*
