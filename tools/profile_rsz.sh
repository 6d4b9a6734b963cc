# Run ON THE GPU BOX (gpurun): the _rsz model's bench line and rocprofv3 kernel statistics of the same command.
#   usage: bash tools/profile_rsz.sh r02   -> gpurun_out/profiles_out/<tag>_rsz_{kernel_stats.csv,bench_line.json,train_step.json}
TAG=${1:-r02}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
export ASR_BENCH_MODEL=mutopia_ccal_cont_rsz
rm -rf $R/gpurun_out/prof_rsz $R/gpurun_out/tune_cache_rsz.txt
export ASR_TUNE_CACHE=$R/gpurun_out/tune_cache_rsz.txt
python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-host-leg --no-secondary --no-dropin > /dev/null 2>&1      # tuner choices made once
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rsz -o s -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-dropin > $R/gpurun_out/prof_rsz.log 2>&1
cd $R
mkdir -p gpurun_out/profiles_out
cp $(find gpurun_out/prof_rsz -name 's_kernel_stats.csv' | head -1) gpurun_out/profiles_out/${TAG}_rsz_kernel_stats.csv
python3 bench.py 2> gpurun_out/bench_rsz.err | tail -1 > gpurun_out/profiles_out/${TAG}_rsz_bench_line.json
python3 tools/bench_secondary.py train 2> /dev/null | tail -1 > gpurun_out/profiles_out/${TAG}_rsz_train_step.json
tail -c 300 gpurun_out/profiles_out/${TAG}_rsz_bench_line.json
