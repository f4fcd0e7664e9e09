# round 6: the fuzzers with new seeds on the build without the store-data hazard
mkdir -p gpurun_out; export TMPDIR=/tmp
for SD in ${SEEDS:-3 4 5}; do
  timeout 900 python tools/fuzz_batched.py 120 $SD > gpurun_out/fuzz_batched_s$SD.log 2>&1
  echo "fuzz_batched seed $SD rc=$? $(grep -a 'configs failed' gpurun_out/fuzz_batched_s$SD.log)"
done
