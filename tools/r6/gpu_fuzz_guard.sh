# round 6: the fuzzers (random image sizes, window sizes, pyramid depths ...) with every device buffer of the library between
# unmapped guard ranges (KVFE_GUARD_ALLOC = $M): an out-of-bounds access of a shape no fixed test has kills the run
mkdir -p gpurun_out; export TMPDIR=/tmp
M=${M:-1}
run() { # name, command...
  local name=$1; shift
  KVFE_GUARD_ALLOC=$M AMD_LOG_LEVEL=1 timeout ${TMO:-900} "$@" > gpurun_out/fg${M}_$name.log 2>&1
  echo "guard $M $name rc=$? | $(grep -a 'configs failed\|mismatching checks' gpurun_out/fg${M}_$name.log | tail -1 | cut -c1-160) | $(grep -a 'Memory access fault' gpurun_out/fg${M}_$name.log | head -1 | cut -c1-120)"
  tail -2 gpurun_out/fg${M}_$name.log | cut -c1-300
}
run components python tools/fuzz_components.py ${NC:-100} ${SD:-91}
run frontend python tools/fuzz_frontend.py ${NF:-80} ${SD:-91}
run variants python tools/fuzz_variants.py ${NV:-40} ${SD:-91}
run batched python tools/fuzz_batched.py ${NB:-50} 1${SD:-91}
