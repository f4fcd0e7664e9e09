# round 4, call d: phase cycle counts of the grouped cornerSubPix kernel (KVFE_SUBPIX_STATS)
mkdir -p gpurun_out; export TMPDIR=/tmp
for leg in none kf_realistic; do
KVFE_SUBPIX_STATS=1 KVFE_SUBPIX_GROUP=1 timeout 300 python bench.py --legs $leg --steps 20 --warmup 5 --repeats 1 --no-stage-events 2>&1 >/dev/null | grep KVFE_SUBPIX_STATS
done
