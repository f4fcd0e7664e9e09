# round 6: tools/fuzz_batched.py seed 2 configuration 79 (intermittent tracking mismatch on streams 8 - 11) with the
# context's own pyramid read back and compared after every step (kvfe_frontend_debug_pyramid)
mkdir -p gpurun_out; export TMPDIR=/tmp
FUZZ_PYR_PROBE=1 FUZZ_REPEAT=${REP:-40} timeout 600 python tools/fuzz_batched.py 120 2 79 > gpurun_out/pyr_probe.log 2>&1
echo "rc=$? $(grep -a 'configs failed' gpurun_out/pyr_probe.log)"
grep -a "PYRAMID" gpurun_out/pyr_probe.log | cut -c1-400 | head -30
grep -a -A2 "MISMATCH array keypoints" gpurun_out/pyr_probe.log | cut -c1-300 | head -40
