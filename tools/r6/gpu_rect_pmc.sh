# round 6: FETCH_SIZE / WRITE_SIZE of rectify_tile_kernel per block order (separate passes)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
cd /tmp
for V in ${VS:-0 2}; do
for C in FETCH_SIZE WRITE_SIZE; do
  KVFE_RECT_XCD=$V timeout 200 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/rp_${V}_$C -o p -- python $R/bench.py --steps 10 --warmup 3 --repeats 1 --legs none --no-stage-events --no-cpu-baseline > $R/gpurun_out/rp.log 2>&1
  echo "RECT_XCD=$V $C rc=$?"
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/rp_${V}_$C -name "*.db" | head -1) 2>&1 | grep -E "^\| kernel|rectify_tile|pyr2|mineig2" | cut -c1-200
done
done
rm -rf $R/gpurun_out/rp_*
