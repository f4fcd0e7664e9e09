# round 4, call z: the first timed region after the warm-up is 10 - 20 % slower since the output transfer is a DMA -- how long
# does that last, and does warming the DMA path when the context is created remove it?
# (result: regions of 20 steps after 8 warm-up steps: 39.2 / 65.0 / 58.1 / 58.2 / 58.8 / 65.5 ... k pairs/s; warming the DMA path at
# kvfe_create changes nothing (38.0 / 65.3 / 58.3 ...); after 60 warm-up steps 58.8 / 64.4 / 58.0 ... -> bench.py runs 40 untimed
# steps in front of every leg's warm-up.  The KVFE_X_DMA_WARM switch is not in the tree.)
mkdir -p gpurun_out; export TMPDIR=/tmp
for W in "" 16; do
KVFE_X_DMA_WARM=$W timeout 300 python bench.py --legs none --steps 20 --warmup 8 --repeats 10 --no-stage-events 2> gpurun_out/z_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[warm=$W]', d['value'], [round(v/1000,1) for v in d['repeats']['values']])"
done
timeout 300 python bench.py --legs none --steps 20 --warmup 60 --repeats 6 --no-stage-events 2> gpurun_out/z_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[warmup 60 steps]', d['value'], [round(v/1000,1) for v in d['repeats']['values']])"
