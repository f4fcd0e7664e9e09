# round 5, call ab: what bounds the chunk loop of the grouped cornerSubPix kernel?  Profiling builds (WRONG results: every
# sum is 0, every corner leaves after one iteration): libkvfe_diag1.so = the chain wave adds nothing, libkvfe_diag2.so = the
# producers produce nothing; chunk-loop cycles per block iteration against the product build
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
for V in libkvfe.so libkvfe_diag1.so libkvfe_diag2.so; do
KVFE_LIB=$L/$V KVFE_SUBPIX_STATS=1 PROBE_N=6 timeout 120 python tools/r5/spg_probe.py > gpurun_out/ab_out.log 2> gpurun_out/ab_err.log
echo "$V: $(tail -1 gpurun_out/ab_out.log | cut -c1-60)"
grep "KVFE_SUBPIX_STATS (group" gpurun_out/ab_out.log gpurun_out/ab_err.log | tail -1
done
