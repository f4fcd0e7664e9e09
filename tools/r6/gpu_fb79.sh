# round 6: the intermittent mismatch of tools/fuzz_batched.py (seed 2, configuration 79) under the library's A/B switches
mkdir -p gpurun_out; export TMPDIR=/tmp
for E in ${ENVS_LIST:-"X=0" "KVFE_LK_IMPL=1" "KVFE_LK4_ORDER=0" "KVFE_PYR2_T2=4"}; do
  env $E FUZZ_REPEAT=${REP:-30} timeout 300 python tools/fuzz_batched.py 120 2 79 > gpurun_out/fb79_ab.log 2>&1
  echo "$E: $(grep -a 'configs failed' gpurun_out/fb79_ab.log) | streams: $(grep -a 'MISMATCH array keypoints$' gpurun_out/fb79_ab.log | awk '{print $5}' | sort | uniq -c | tr '\n' ' ')"
done
