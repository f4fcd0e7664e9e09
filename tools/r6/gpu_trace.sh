# round 6: kernel trace of the headline step (single HIP stream so that durations are the kernels' own), top rows
# usage: [ENVS="A=1 B=2"] bash tools/r6/gpu_trace.sh <tag> [grep pattern]
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
tag=${1:-t}; pat=${2:-.}
cd /tmp
env $ENVS timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tr_$tag -o s -- python $R/bench.py --steps 12 --warmup 4 --repeats 1 --legs none --no-stage-events --no-cpu-baseline ${SINGLE:+--single-hip-stream} > $R/gpurun_out/tr_$tag.log 2>&1; echo "rc=$?"
db=$(find $R/gpurun_out/tr_$tag -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db > $R/gpurun_out/tr_$tag.md
head -2 $R/gpurun_out/tr_$tag.md; grep -E "$pat" $R/gpurun_out/tr_$tag.md | head -${N:-12}
rm -rf $R/gpurun_out/tr_$tag
