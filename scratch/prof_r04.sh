#!/bin/bash
# Round-4 profile collection (run on the GPU box through gpurun, from the repo root); summaries land in gpurun_out/r04/.
# PMC passes are separate runs with --kernel-trace only (never combined with other trace domains).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B="python3 bench.py --no-extra-workloads --no-cpu-baseline --steps 3 --warmup 1"
stats() {   # name, extra args
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$1 -o t -- $B $2 > $O/$1.log 2>&1
  cp $O/raw_$1/t_kernel_stats.csv $O/r04_$1_kernel_stats.csv
}
pmc() {     # name, counters..., then -- extra args
  local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $O/raw_$name -o t -- $B "$@" > $O/$name.log 2>&1
  python3 scratch/pmc_summary.py $O/raw_$name/t_counter_collection.csv > $O/r04_$name.csv
}
stats xe_b256_bf16 ""
stats decode_stack_b1024_beam5 "--workload decode"
stats sparse_decode_stack_b1024_beam5 "--workload sparse_decode"
stats scst_b256_ns5 "--workload scst"
stats scst_nodrop_b256_ns5 "--workload scst --variant nodrop"
stats sparse_xe_988_kernels "--workload sparse_xe --variant 988_kernels"
pmc xe_b256_pmc_fetch_size FETCH_SIZE --
pmc xe_b256_pmc_write_size WRITE_SIZE --
pmc xe_b256_pmc_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --
pmc decode_stack_pmc_fetch_size FETCH_SIZE -- --workload decode
pmc decode_stack_pmc_write_size WRITE_SIZE -- --workload decode
pmc sparse_decode_stack_pmc_fetch_size FETCH_SIZE -- --workload sparse_decode
pmc sparse_decode_stack_pmc_write_size WRITE_SIZE -- --workload sparse_decode
pmc scst_pmc_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- --workload scst
rm -rf $O/raw_*
python3 bench.py --steps 50 --warmup 10 > $O/r04_default_bench_line.json 2> $O/bench.err
wc -c $O/r04_default_bench_line.json; tail -2 $O/bench.err
ls $O | head -40
