# Where a wave of lk_kernel_sys spends its time, and the time line of one launch (profiles/r2_v5_lk_analysis.md, 3).
# Needs an instrumented build of the library next to the product (never the product itself):
#   cd kimera_vio_amd/csrc; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
#   hipcc $F -DKVFE_LK_PROF=1 -c k_track.hip -o /tmp/k_track_prof.o      # 1: cycle stamps per phase, 2: start / end only
#   hipcc --offload-arch=gfx950 -shared -fPIC -o libkvfe_prof.so k_rectify.o k_detect.o /tmp/k_track_prof.o k_stereo.o \
#         k_ransac.o k_dense.o k_components.o kvfe_api.o host_calib.o host_input.o -lz
# usage: gpurun --timeout 300 -- 'bash tools/gpu_lk_prof.sh'; the report is printed at exit on stderr (KVFE_LK_PROF ...)
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/kimera_vio_amd/csrc
for lib in libkvfe_prof.so; do
  echo "== $lib, batched leg"; KVFE_LIB=$L/$lib python bench.py --legs none --steps 10 --warmup 3 --repeats 1 --no-stage-events 2>&1 | grep KVFE_LK_PROF
  echo "== $lib, single stream"; KVFE_LIB=$L/$lib python bench.py --config c2 --legs none --steps 60 --warmup 5 --repeats 1 --no-stage-events 2>&1 | grep KVFE_LK_PROF
done
