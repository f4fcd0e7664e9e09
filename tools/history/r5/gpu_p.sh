# round 5, call p: the rows' own speed -- a build in which nobody waits (wrong results, timing only)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
for n in 8 1; do
KVFE_X_AGP_MIN=1 KVFE_LIB=$L/libkvfe_agfree.so timeout 120 python tools/r5/dense_probe.py $n 2>&1 | grep -v amdgpu.ids

done
