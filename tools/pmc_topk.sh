# PMC counters of the top-k filter kernel (diagnostics)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/pmc_topk1 $R/gpurun_out/pmc_topk2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_topk1 -o p -- python3 $R/tools/topk_one.py ${1:-1024} 25 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_topk2 -o p -- python3 $R/tools/topk_one.py ${1:-1024} 25 > /dev/null 2>&1
