# HBM traffic of every kernel (MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE in separate passes;
# FETCH_SIZE x2 on gfx950 for wide coalesced reads).  Output: gpurun_out/traffic_{rd,wr}/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
export ASR_SINGLE_STREAM=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/traffic_rd -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic_rd.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic_wr -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic_wr.log 2>&1
ls $R/gpurun_out/traffic_rd $R/gpurun_out/traffic_wr
