# round 6: the store-data hazard of pyr2_kernel's level-0 copy -- the regression tests on a build WITH the hazard
# (kimera_vio_amd/csrc/libkvfe_oldpyr.so = k_rectify.hip of commit fce4998 linked with the current objects; the in-place
# test is expected to fail) and on the current build, then the fuzz configuration that found it, 40 times, with the
# pyramid read back
mkdir -p gpurun_out; export TMPDIR=/tmp
T="tests/test_gpu_r6_regressions.py -k level0_copy"
if [ -f kimera_vio_amd/csrc/libkvfe_oldpyr.so ]; then
KVFE_LIB=$PWD/kimera_vio_amd/csrc/libkvfe_oldpyr.so timeout 600 python -m pytest $T -q > gpurun_out/hazard_old.log 2>&1
echo "old build rc=$? $(tail -1 gpurun_out/hazard_old.log) | $(grep -a 'AssertionError: ' gpurun_out/hazard_old.log | head -2 | cut -c1-300)"
fi
timeout 600 python -m pytest $T -q > gpurun_out/hazard_new.log 2>&1
echo "current build rc=$? $(tail -1 gpurun_out/hazard_new.log)"
[ -n "$PROBE" ] && bash tools/r6/gpu_pyr_probe.sh
