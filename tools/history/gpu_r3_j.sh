mkdir -p gpurun_out; export TMPDIR=/tmp
KVFE_SELECT_PROF=1 timeout 200 python bench.py --config c2 --steps 100 --warmup 10 --repeats 1 --legs none --no-stage-events > gpurun_out/j_sel.json 2> gpurun_out/j_sel.err; grep KVFE_SELECT_PROF gpurun_out/j_sel.err
KVFE_SELECT_PROF=1 timeout 200 python bench.py --steps 20 --warmup 5 --repeats 1 --legs kf_realistic --no-stage-events > gpurun_out/j_sel2.json 2> gpurun_out/j_sel2.err; grep KVFE_SELECT_PROF gpurun_out/j_sel2.err
