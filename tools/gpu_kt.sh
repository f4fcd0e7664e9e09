# rocprofv3 kernel trace of the batched leg alone (the per-kernel durations behind the bench line of the same build)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
cd $R; python tools/rocpd_stats.py gpurun_out/prof_kt/kt_results.db | head -8
