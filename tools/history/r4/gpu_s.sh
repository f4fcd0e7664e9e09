# round 4, call s: where do the 0.19 - 0.26 ms of the one-corner-per-block cornerSubPix launch go (11 corners per stream)?
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
KVFE_SUBPIX_STATS=1 timeout 300 python bench.py --legs none --steps 20 --warmup 5 --repeats 1 --single-hip-stream 2> gpurun_out/s_stats.err > /dev/null; grep KVFE_SUBPIX gpurun_out/s_stats.err | cut -c1-300
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_s -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none --single-hip-stream --no-stage-events > $R/gpurun_out/prof_s.log 2>&1; echo "kt rc=$?"
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/prof_s -name "*.db" | head -1) | head -24 | cut -c1-150
