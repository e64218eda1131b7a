#!/bin/bash
# counters of the grouped weight-gradient kernel on one layer shape: bash scratch/wgrad_pmc.sh dec 4
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wgpmc; mkdir -p $O; : > $O/wgrad_pmc_$1_$2.txt
for ctr in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/raw -o t -- python3 scratch/wgrad_group_one.py $1 $2 5 > $O/log.txt 2>&1
  python3 scratch/pmc_kernels.py $O/raw/t_counter_collection.csv wgrad_group >> $O/wgrad_pmc_$1_$2.txt 2>&1
  rm -rf $O/raw
done
cat $O/wgrad_pmc_$1_$2.txt
