# round 5, call j: does the enqueue of a pinned 23 MB H2D copy block the host?  (tools/r5/h2d_probe.py)
export TMPDIR=/tmp
timeout 120 python tools/r5/h2d_probe.py 2>&1 | grep -v amdgpu.ids
HIP_LAUNCH_BLOCKING=0 AMD_DIRECT_DISPATCH=0 timeout 120 python tools/r5/h2d_probe.py 2>&1 | grep -v amdgpu.ids | sed 's/^/[AMD_DIRECT_DISPATCH=0] /'
