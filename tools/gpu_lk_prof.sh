# timeline of one lk_kernel_sys launch (-DKVFE_LK_PROF builds, built by hand: see DESIGN.md 4.3)
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/kimera_vio_amd/csrc
for lib in libkvfe_baseprof.so libkvfe_prof.so libkvfe_prof6.so; do
  echo "== $lib c3"; KVFE_LIB=$L/$lib python bench.py --legs none --steps 10 --warmup 3 --repeats 1 --no-stage-events 2>&1 | grep "KVFE_LK_PROF one\|timeline"
done
