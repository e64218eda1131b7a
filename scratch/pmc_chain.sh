#!/bin/bash
# counters of the two forms of the forward chain kernel (scratch/chain_bench.py): separate --pmc pass, kernel trace only
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c; mkdir -p $O
for mode in wide narrow; do
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/raw_$mode -o t -- python3 scratch/chain_bench.py $mode > $O/$mode.log 2>&1
echo "$mode rc=$?"
python3 scratch/pmc_summary.py $O/raw_$mode/t_counter_collection.csv | grep -E "row_chain|kernel,counter" > $O/r04_chain_${mode}_pmc.csv
cat $O/r04_chain_${mode}_pmc.csv
done
rm -rf $O/raw_*
