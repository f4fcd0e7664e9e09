#!/bin/bash
# tools/save_profiles.sh <tag>: summarise the rocprofv3 databases of the last tools/gpu_profile.sh run
# (gpurun_out/, scratch) into profiles/<tag>_*.{md,json,log} (tracked).
set -e
T=$1
cp gpurun_out/bench.json profiles/${T}_bench.json
cp gpurun_out/gpu_tests.log profiles/${T}_gpu_tests.log
{ echo "# rocprofv3 --kernel-trace --stats, python bench.py --steps 20 --warmup 5 (64 streams, KF every frame)"; echo;
  python tools/rocpd_stats.py gpurun_out/prof_kt/kt_results.db; } > profiles/${T}_rocprof_kernel_stats.md
{ echo "# rocprofv3 --pmc (separate passes; kernels serialised), same command.  FETCH_SIZE / WRITE_SIZE in KB per launch"; echo;
  python tools/rocpd_pmc.py gpurun_out/prof_fetch/f_results.db gpurun_out/prof_write/w_results.db; echo;
  echo "SQ pass 1 (SQ_* cycle counters are in quad-cycles summed over all SIMDs/XCDs):"; echo;
  python tools/rocpd_pmc.py gpurun_out/prof_sq/s_results.db 2>/dev/null || echo "(not collected in this run)"; echo;
  echo "SQ pass 2:"; echo;
  python tools/rocpd_pmc.py gpurun_out/prof_sq2/s2_results.db 2>/dev/null || echo "(not collected in this run)"; } > profiles/${T}_pmc.md
ls -la profiles/${T}_*
python tools/rocpd_traffic.py gpurun_out/prof_fetch/f_results.db gpurun_out/prof_write/w_results.db > profiles/${T}_pmc_traffic.json
cp profiles/${T}_pmc_traffic.json profiles/pmc_traffic_latest.json
