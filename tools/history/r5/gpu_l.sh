# round 5, call l: where the two-pass aggregation's time goes (profiling build: row durations, wait turns)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
for n in 8 4; do
timeout 120 python tools/r5/dense_probe.py $n 2>&1 | grep -v amdgpu.ids
KVFE_LIB=$L/libkvfe_agprof.so timeout 120 python tools/r5/dense_probe.py $n 2>&1 | grep -v amdgpu.ids
done
KVFE_X_DENSE8=1 timeout 120 python tools/r5/dense_probe.py 8 2>&1 | grep -v amdgpu.ids
