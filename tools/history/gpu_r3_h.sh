# round 3, call h: rectification with tabulated source boxes (default) / boxes recomputed per block (KVFE_RECT_NO_BOX=1) /
# both LDS buffers requested together (KVFE_RECT_PAIRS=1): parity suites, bench stage times on one box, and the kernel
# alone (KVFE_NO_SIDE_STREAM=1: nothing runs beside it) from a kernel trace per variant.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
TESTS="tests/test_gpu_parity.py tests/test_gpu_components_r2.py tests/test_gpu_bench_configs.py tests/test_gpu_fuzz_slices.py tests/test_gpu_pins_r2.py"
timeout 500 python -m pytest $TESTS -m gpu -x -q > gpurun_out/h_tests_box.log 2>&1; echo "tests (boxes tabulated) rc=$?"; tail -2 gpurun_out/h_tests_box.log
KVFE_RECT_PAIRS=1 timeout 500 python -m pytest $TESTS -m gpu -x -q > gpurun_out/h_tests_pairs.log 2>&1; echo "tests (pairs) rc=$?"; tail -2 gpurun_out/h_tests_pairs.log
run() {
env $1 timeout 300 python bench.py --legs none --steps 30 --warmup 8 --repeats 2 2> gpurun_out/h_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items()))"
}
for rep in 1 2; do
  run KVFE_RECT_NO_BOX=1
  run KVFE_X=0
  run KVFE_RECT_PAIRS=1
  run "KVFE_RECT_PAIRS=1 KVFE_RECT_TILE_MODE=1"
done
cd /tmp
for v in KVFE_RECT_NO_BOX=1 KVFE_X=0 KVFE_RECT_PAIRS=1 "KVFE_RECT_PAIRS=1 KVFE_RECT_TILE_MODE=1" "KVFE_RECT_TILE_MODE=1"; do
  n=$(echo "$v" | tr ' =' '__')
  env $v KVFE_NO_SIDE_STREAM=1 timeout 120 rocprofv3 --kernel-trace -d $R/gpurun_out/h_kt_$n -o kt -- python $R/bench.py --steps 12 --warmup 4 --repeats 1 --legs none --no-stage-events > $R/gpurun_out/h_kt.log 2>&1
  echo "[$v] alone:"; python $R/tools/rocpd_stats.py $(find $R/gpurun_out/h_kt_$n -name "*.db" | head -1) | grep -E "rectify_tile|stereo_match_kernel|subpix"
done
