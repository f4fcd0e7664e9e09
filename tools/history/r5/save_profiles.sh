#!/bin/bash
# tools/r5/save_profiles.sh <tag>: summarise the rocprofv3 databases of the last tools/gpu_r3_round.sh PROFILE=1 run
# (gpurun_out/, scratch) into profiles/<tag>_*.{md,json,log} (tracked).
set -e
T=$1
cp gpurun_out/bench.json profiles/${T}_bench.json; cp gpurun_out/bench_line.json profiles/${T}_bench_line.json
cp gpurun_out/gpu_tests.log profiles/${T}_gpu_tests.log
for L in "" _c5 _c2 _e; do
  db=$(find gpurun_out/prof_kt$L -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then
  { echo "# rocprofv3 --kernel-trace --stats, bench.py leg ${L:-_c3} alone (see tools/gpu_r3_round.sh for the command)"; echo;
    python tools/rocpd_stats.py $db; } > profiles/${T}_rocprof_kernel_stats${L}.md
  fi
done
ARGS=""
for L in c3_kf c5_kf dense dense_c5; do
  f=$(find gpurun_out/pmc_${L}_FETCH_SIZE -name "*.db" 2>/dev/null | head -1); w=$(find gpurun_out/pmc_${L}_WRITE_SIZE -name "*.db" 2>/dev/null | head -1)
  if [ -n "$f" ] && [ -n "$w" ]; then ARGS="$ARGS --leg $L $f $w"; fi
done
python tools/rocpd_traffic.py $ARGS > profiles/${T}_pmc_traffic.json
cp profiles/${T}_pmc_traffic.json profiles/pmc_traffic_latest.json
s1=$(find gpurun_out/prof_sq -name "*.db" 2>/dev/null | head -1); s2=$(find gpurun_out/prof_sq2 -name "*.db" 2>/dev/null | head -1)
if [ -n "$s1" ]; then
{ echo "# rocprofv3 --pmc SQ passes, bench.py c3 leg alone (SQ_* cycle counters are quad-cycles summed over all SIMDs/XCDs)"; echo;
  python tools/rocpd_pmc.py $s1; echo;
  [ -n "$s2" ] && python tools/rocpd_pmc.py $s2; } > profiles/${T}_pmc_sq.md
python tools/rocpd_pmc.py --json $s1 > profiles/pmc_valu_latest.json
fi
ls -la profiles/${T}_*
