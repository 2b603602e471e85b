R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for V in 102 302; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/mfma_$V -o p -- python3 $R/tools/pmc_probe.py 0 4096 $V > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/lds_$V -o p -- python3 $R/tools/pmc_probe.py 0 4096 $V > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/grbm_$V -o p -- python3 $R/tools/pmc_probe.py 0 4096 $V > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$V -o p -- python3 $R/tools/pmc_probe.py 0 4096 $V > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/act_$V -o p -- python3 $R/tools/pmc_probe.py 0 4096 $V > /dev/null 2>&1
done
