#!/bin/bash
# Round-6 profile collection (run on the GPU box through gpurun, from the repo root):
#     gpurun -- "GIT_HEAD=$(git rev-parse HEAD) bash scratch/prof_r06.sh stats|pmc|final|sweep"
# four calls (the whole set does not fit one call's time limit): `stats` = kernel-trace summaries of every workload, `pmc` = the HBM
# counter passes, `sweep` = host-thread sweep of the CPU baseline (40 min: the 256-thread setting), `final` = the default bench line + the A/B texts (run it after the pmc call's CSVs have been
# copied into profiles/: bench.py reads them for roofline.traffic).
# Summaries land in gpurun_out/r06/ together with r06_manifest.json = {lib_md5, git_head, ...}: the identity of the library every
# pass ran.  bench.py's pmc_traffic() refuses the PMC passes when that md5 is not the md5 of the library it has loaded.
# PMC passes are separate runs with --kernel-trace only (never combined with other trace domains).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
MD5=$(md5sum sparse-image-captioning_amd/libortk.so | cut -d' ' -f1)
B="python3 bench.py --no-extra-workloads --no-cpu-baseline --steps 3 --warmup 1"
stats() {   # name, extra args
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$1 -o t -- $B $2 > $O/$1.log 2>&1
  cp $O/raw_$1/t_kernel_stats.csv $O/r06_$1_kernel_stats.csv
}
pmc() {     # name, counter, then -- extra args
  local name=$1; shift; local ctr=$1; shift; shift
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/raw_$name -o t -- $B "$@" > $O/$name.log 2>&1
  python3 scratch/pmc_summary.py $O/raw_$name/t_counter_collection.csv > $O/r06_$name.csv
}
PART=${1:-stats}
if [ "$PART" = stats ]; then
stats xe_b256_bf16 ""
stats decode_stack_b1024_beam5 "--workload decode"
stats sparse_decode_stack_b1024_beam5 "--workload sparse_decode"
stats scst_b256_ns5 "--workload scst"
stats sparse_xe_b256 "--workload sparse_xe"
stats sparse_xe_kernels_b256 "--workload sparse_xe --variant kernels"
stats decode_fp32_b1024_beam5 "--workload decode --variant fp32"
fi
if [ "$PART" = pmc ]; then
for c in fetch_size write_size; do
  C=$(echo $c | tr a-z A-Z)
  pmc xe_b256_pmc_$c $C --
  pmc decode_stack_pmc_$c $C -- --workload decode
  pmc sparse_decode_stack_pmc_$c $C -- --workload sparse_decode
  pmc sparse_decode_988_stack_pmc_$c $C -- --workload sparse_decode --variant 988
  pmc scst_b256_pmc_$c $C -- --workload scst
  pmc sparse_xe_b256_pmc_$c $C -- --workload sparse_xe
  pmc sparse_xe_kernels_b256_pmc_$c $C -- --workload sparse_xe --variant kernels
done
fi
rm -rf $O/raw_*
python3 - <<PY
import json, time
json.dump({"lib_md5": "$MD5", "git_head": "${GIT_HEAD:-unknown}", "collected_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
           "command": "scratch/prof_r06.sh $PART", "bench": "$B",
           "note": "every r06_*_kernel_stats.csv and r06_*_pmc_*.csv of round 6 was produced by the library with this md5 (three gpurun calls: stats, pmc, final)"},
          open("$O/r06_manifest_$PART.json", "w"), indent=1)
PY
if [ "$PART" = sweep ]; then     # (its own call: the 128- and 256-thread settings run for minutes per step)
python3 scratch/cpu_thread_sweep.py > $O/r06_cpu_thread_sweep.txt 2>&1
fi
if [ "$PART" = final ]; then
python3 bench.py --steps 50 --warmup 10 > $O/r06_default_bench_line.json 2> $O/bench.err
wc -c $O/r06_default_bench_line.json; tail -2 $O/bench.err
# the round's A/B and kernel-alone figures cited from DESIGN.md / the sources, from this library
python3 scratch/wgrad_group_ab.py > $O/r06_wgrad_group_ab.txt 2>&1
python3 scratch/wgrad_group_ab.py gemm_epilogue=1 gemm_epilogue=0 ln_fuse=8 ln_fuse=0 wgrad_group_wgs=64 wgrad_group_wgs=96 ln_fuse=1 gemm_impl=2 gemm_impl=3 > $O/r06_xe_step_switches_ab.txt 2>&1
python3 scratch/wgrad_group_bench.py 0 > $O/r06_wgrad_group.txt 2>&1
python3 scratch/box_bwd_bench.py > $O/r06_box_bwd.txt 2>&1
python3 scratch/gemm_fixed_cost.py > $O/r06_gemm_fixed_cost.txt 2>&1
python3 scratch/ln_bench.py > $O/r06_ln_bwd.txt 2>&1
python3 scratch/gemm_ln_bench.py > $O/r06_gemm_ln_fused.txt 2>&1
./scratch/micro/store_like_gemm > $O/r06_store_pattern.txt 2>&1
python3 scratch/scst_tuning_ab.py samp_epilogue=0 samp_epilogue=1 gemm_epilogue=1 ln_fuse=8 > $O/r06_scst_switches_ab.txt 2>&1
python3 scratch/gy_bf16_grad_error.py > $O/r06_ln_gradient_bf16.txt 2>&1
fi
echo "lib md5 $MD5"; ls $O | head -80
