# Run ON THE GPU BOX (gpurun): refreshes every measurement the round's profiles/ are built from.
#   1. rocprofv3 --kernel-trace --stats of the default bench command          -> gpurun_out/prof_stats/
#   2. HBM traffic counters, separate passes (tools/pmc_traffic.sh)           -> gpurun_out/traffic_{rd,wr}/
#   3. secondary benchmarks (train step, CCA fit, top-k, rank)                -> gpurun_out/secondary.jsonl
#   4. the bench line itself (with the CPU baseline leg)                      -> gpurun_out/bench_line.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/traffic_rd $R/gpurun_out/traffic_wr $R/gpurun_out/tune_cache.txt
# the autotuner's choices are made once and re-used by every run below (ASR_TUNE_CACHE), so that the profiled
# processes contain the steady-state launches only and all runs execute the same kernels
export ASR_TUNE_CACHE=$R/gpurun_out/tune_cache.txt
python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o s -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
bash $R/tools/pmc_traffic.sh > /dev/null 2>&1
cd $R
python3 tools/summarize_profiles.py r01 --traffic-only    # bench.py reads the per-symbol HBM traffic from profiles/
python3 tools/bench_secondary.py > gpurun_out/secondary.jsonl 2> gpurun_out/secondary.err
python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
tail -c 600 gpurun_out/bench_line.json; ls gpurun_out/prof_stats gpurun_out/traffic_rd gpurun_out/traffic_wr
