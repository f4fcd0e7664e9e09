# round 4, call y: why is the staged (PCIe-inclusive) rate 32 k now and was 43.6 k in round 3?
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
L=$PWD/kimera_vio_amd/csrc
for X in 0 1 0 1; do
KVFE_X_SERIAL_STAGED=$X timeout 200 python tools/r4/staged_probe.py 2>&1 | grep staged | sed "s/^/[serial=$X] /"
done
KVFE_X_SERIAL_STAGED=1 timeout 600 python -m pytest tests/test_gpu_pipelined_r3.py tests/test_input_side.py -m gpu -q -x 2>&1 | tail -3
for X in 0 1; do
KVFE_X_SERIAL_STAGED=$X timeout 300 python bench.py --legs pcie --steps 10 --warmup 4 --repeats 1 2> gpurun_out/y_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[serial=$X]', d['value'], d.get('pcie_inclusive'))"
done
