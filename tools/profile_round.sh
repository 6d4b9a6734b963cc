# Run ON THE GPU BOX (gpurun): every measurement the round's profiles/ are built from, all on the same tuner choices.
#   usage: bash tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of the default bench command                   -> gpurun_out/prof_stats/
#   2. PMC passes (separate processes, --kernel-trace only - never combined with other trace domains):
#        FETCH_SIZE | WRITE_SIZE | SQ MFMA/VALU/wait counters + GRBM_GUI_ACTIVE | LDS/VMEM counters
#      each with ASR_LAUNCH_LOG so that every dispatch can be attributed to its layer   -> gpurun_out/pmc_*/
#   3. tools/summarize_pmc.py <tag>  -> profiles/<tag>_{kernel_stats.csv,hbm_traffic_by_symbol.json,
#                                        mfma_busy_by_symbol.json,pmc_by_layer.csv}
#   4. secondary benchmarks and the bench line itself (reads the fresh profiles)        -> profiles/<tag>_*.json(l)
TAG=${1:-r03}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc_rd $R/gpurun_out/pmc_wr $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_lds $R/gpurun_out/tune_cache.txt $R/gpurun_out/launch_*.log
export ASR_TUNE_CACHE=$R/gpurun_out/tune_cache.txt
python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-host-leg --no-secondary --no-dropin > /dev/null 2>&1      # tuner choices made once
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-dropin > $R/gpurun_out/prof_stats.log 2>&1
B="python3 $R/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-host-leg --no-secondary --no-dropin --profile-all"
ASR_LAUNCH_LOG=$R/gpurun_out/launch_rd.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_rd -o p -- $B > $R/gpurun_out/pmc_rd.log 2>&1
ASR_LAUNCH_LOG=$R/gpurun_out/launch_wr.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_wr -o p -- $B > $R/gpurun_out/pmc_wr.log 2>&1
ASR_LAUNCH_LOG=$R/gpurun_out/launch_sq.log rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -o p -- $B > $R/gpurun_out/pmc_sq.log 2>&1
ASR_LAUNCH_LOG=$R/gpurun_out/launch_lds.log rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_lds -o p -- $B > $R/gpurun_out/pmc_lds.log 2>&1
cd $R
python3 tools/summarize_pmc.py $TAG > gpurun_out/summarize_pmc.log 2>&1; tail -3 gpurun_out/summarize_pmc.log
python3 tools/bench_secondary.py > gpurun_out/secondary.jsonl 2> gpurun_out/secondary.err
python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
cp gpurun_out/secondary.jsonl profiles/${TAG}_secondary_bench.jsonl; tail -1 gpurun_out/bench_line.json > profiles/${TAG}_bench_line.json
mkdir -p gpurun_out/profiles_out; cp profiles/${TAG}_* gpurun_out/profiles_out/
tail -c 400 gpurun_out/bench_line.json
