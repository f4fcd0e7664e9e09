# A/B of two builds of libkvfe.so on ONE GPU box (box-to-box speed differs by +-8 %): parity tests that exercise LK
# on build B, then the batched bench leg alternating A, B, A, B.  KVFE_LIB selects the build (kimera_vio_amd/lib.py).
#   A = kimera_vio_amd/csrc/libkvfe_base.so : build it from the commit to compare against, e.g.
#         git stash; make -C kimera_vio_amd/csrc -j8; cp kimera_vio_amd/csrc/libkvfe.so kimera_vio_amd/csrc/libkvfe_base.so
#         git stash pop; touch kimera_vio_amd/csrc/*.hip; make -C kimera_vio_amd/csrc -j8
#       (*.so files are git-ignored but travel to the GPU box with the snapshot)
#   B = kimera_vio_amd/csrc/libkvfe.so (the working tree)
# usage: gpurun --timeout 600 -- 'bash tools/gpu_lk_ab.sh'        (profiles/r2_v5_lk_analysis.md, table 1)
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/kimera_vio_amd/csrc
python -m pytest tests -m gpu -x -q -k "track or lk or optical or sequence or bench_configs or fuzz or adapter" 2>&1 | tail -2
run() {
KVFE_LIB=$L/$1 python bench.py --legs none --steps 30 --warmup 8 --repeats 2 --stage-event-stride 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step_summed_over_groups']; print('$1 c3', d['value'], 'lk', s.get('lk_track'), s)"
}
for lib in libkvfe_base.so libkvfe.so libkvfe_base.so libkvfe.so; do [ -f $L/$lib ] && run $lib; done
