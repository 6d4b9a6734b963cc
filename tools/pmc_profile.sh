export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
export ASR_SINGLE_STREAM=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc1 -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk 250 > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc2 -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk 250 > $R/gpurun_out/pmc2.log 2>&1
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --chunk 250 2>&1 | tail -1 | cut -c 1-200
ls $R/gpurun_out/pmc1 $R/gpurun_out/pmc2
