# rocprofv3 kernel trace of one bench leg alone (the per-kernel durations behind the bench line of the same build)
# usage: gpurun -- 'bash tools/gpu_kt.sh [c3|c2|c5]'
CFG=${1:-c3}
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
STEPS="--steps 20 --warmup 5"; [ "$CFG" = c2 ] && STEPS="--steps 200 --warmup 20 --no-stage-events"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_$CFG -o kt -- python $R/bench.py --config $CFG $STEPS --repeats 1 --legs none > $R/gpurun_out/prof_kt_$CFG.log 2>&1; echo "kt rc=$?"
cd $R; python tools/rocpd_stats.py gpurun_out/prof_kt_$CFG/kt_results.db | head -8
