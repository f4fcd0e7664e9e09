# round 5, call q: how many blocks of the two-pass aggregation a CU holds
export TMPDIR=/tmp
KVFE_X_AGP_OCC=1 timeout 120 python tools/r5/dense_probe.py 8 2>&1 | grep -v amdgpu.ids | sort | uniq -c | head
