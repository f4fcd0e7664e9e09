# round 6: the fuzzers with new seeds on the build without the store-data hazard.  SEEDS: fuzz_batched seeds (>= 100 draw the
# second campaign's extensions: 3 / 4 / 32 / 40 streams, klt_max_level 0, equalizeImage, stream groups, 1280 x 720);
# OTHERS=1: the four older fuzzers with seeds 81 - 84
mkdir -p gpurun_out; export TMPDIR=/tmp
for SD in ${SEEDS:-101 102}; do
  timeout ${TMO:-1200} python tools/fuzz_batched.py ${NCFG:-120} $SD > gpurun_out/fuzz_batched_s$SD.log 2>&1
  echo "fuzz_batched seed $SD rc=$? $(grep -a 'configs failed' gpurun_out/fuzz_batched_s$SD.log) | $(grep -ac ' ok ' gpurun_out/fuzz_batched_s$SD.log) ok, $(grep -ac 'create refused' gpurun_out/fuzz_batched_s$SD.log) refused"
  grep -a "FAILED\|DEVICE ERROR\|Traceback" gpurun_out/fuzz_batched_s$SD.log | head -5 | cut -c1-400
done
if [ -n "$OTHERS" ]; then
  timeout 900 python tools/fuzz_frontend.py 120 81 > gpurun_out/fuzz_fe_81.log 2>&1; echo "fuzz_frontend rc=$? $(tail -1 gpurun_out/fuzz_fe_81.log | cut -c1-200)"
  timeout 900 python tools/fuzz_variants.py 60 82 > gpurun_out/fuzz_var_82.log 2>&1; echo "fuzz_variants rc=$? $(tail -1 gpurun_out/fuzz_var_82.log | cut -c1-200)"
  timeout 900 python tools/fuzz_components.py 120 83 > gpurun_out/fuzz_comp_83.log 2>&1; echo "fuzz_components rc=$? $(tail -1 gpurun_out/fuzz_comp_83.log | cut -c1-200)"
  timeout 600 python tools/fuzz_ransac.py 40 84 > gpurun_out/fuzz_ransac_84.log 2>&1; echo "fuzz_ransac rc=$? $(tail -1 gpurun_out/fuzz_ransac_84.log | cut -c1-200)"
fi
