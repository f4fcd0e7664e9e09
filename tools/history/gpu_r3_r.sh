mkdir -p gpurun_out; export TMPDIR=/tmp
KVFE_SUBPIX_STATS=1 timeout 200 python bench.py --steps 30 --warmup 8 --repeats 1 --legs none --no-stage-events > gpurun_out/r_stats.json 2> gpurun_out/r_stats.err; grep KVFE_SUBPIX_STATS gpurun_out/r_stats.err
KVFE_SUBPIX_STATS=1 timeout 200 python bench.py --config c2 --steps 100 --warmup 8 --repeats 1 --legs none --no-stage-events > gpurun_out/r_stats2.json 2> gpurun_out/r_stats2.err; grep KVFE_SUBPIX_STATS gpurun_out/r_stats2.err
