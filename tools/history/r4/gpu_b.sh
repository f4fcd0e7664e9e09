# round 4, call b: SQ counters of the dense kernels, min-eigenvalue kernel with and without the run walk
VARIANTS='KVFE_MINEIG_SKIP=0|KVFE_MINEIG_SKIP=1' KERNELS='mineig|rectify|pyr2|subpix|lk_kernel|select' bash tools/gpu_pmc_kernel.sh 2>&1 | cut -c1-260
