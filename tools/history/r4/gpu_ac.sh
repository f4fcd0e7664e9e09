# round 4, call ac: where a wave of mineig2_kernel spends its life (variant builds with -DKVFE_ME_PROF: cycle stamps per wave
# of the last launch), strips of 80 (default) / 60 / 40 rows
# (result, 80 / 60 / 40 rows: mean wave 73 / 69 / 65 k cycles -- mask phase 26 / 23 / 24 k, row loop 43 / 41 / 36 k -- slowest wave 175 / 154 / 155 k,
# launch 0.079 / 0.083 / 0.092 ms; pixel rows needed 25 of 85.  Build the variants with -DKVFE_ME_PROF [-DKVFE_ME_ROWS_OVERRIDE=n] on k_detect.hip)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
for V in libkvfe_meprof.so libkvfe_meprof60.so libkvfe_meprof40.so; do
KVFE_LIB=$L/$V timeout 300 python bench.py --legs none --steps 20 --warmup 5 --repeats 1 --stage-event-stride 2 2> gpurun_out/ac.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('[$V] mineig %.4f ms, step %.4f' % (st['mineig_localmax'], d['ms_per_step']))"
grep KVFE_ME_PROF gpurun_out/ac.err
done
