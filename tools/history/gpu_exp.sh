export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_c5 -o kt -- python $R/bench.py --width 1280 --height 720 --features 1000 --klt-max-level 3 --batch 32 --steps 12 --warmup 4 --no-cpu-baseline --no-single-stream > $R/gpurun_out/prof_kt_c5.log 2>&1; echo "kt rc=$?"
cd $R
