#!/bin/bash
# Round profile set on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02 "nsff_static_1024x128 nsff_zest_val_1024x128" "bf16 f16x3"
# Per workload and operand type: rocprofv3 --kernel-trace --stats --output-format csv of the bench command
# (one mode per run: --no-modes), then separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy/MFMA counters;
# never together with a trace domain).  Summaries land in gpurun_out/profiles_<tag>/; copy them into
# profiles/ afterwards (tools/pmc_traffic.py turns the two HBM passes into bytes per launch).
set -e
tag=$1; wls=$2; modes=${3:-bf16}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/profiles_$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for wl in $wls; do
 for m in $modes; do
  n=${wl}_${m}
  cmd="python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-modes --mode $m --workload $wl"
  d=/tmp/prof_$n; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/ks -o ks -- $cmd > $out/${tag}_${n}_bench_under_rocprofv3.json 2> $out/${n}_ks.err
  cp $(find $d/ks -name "*kernel_stats.csv" | head -1) $out/${tag}_${n}_kernel_stats.csv
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d $d/f -o f -- $cmd > /dev/null 2>&1
  grep -E "Counter_Name|fused_blocks_kernel" $(find $d/f -name "*counter_collection.csv" | head -1) > $out/${tag}_${n}_pmc_fetch.csv
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d $d/w -o w -- $cmd > /dev/null 2>&1
  grep -E "Counter_Name|fused_blocks_kernel" $(find $d/w -name "*counter_collection.csv" | head -1) > $out/${tag}_${n}_pmc_write.csv
  rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $d/s -o s -- $cmd > /dev/null 2>&1
  grep -E "Counter_Name|fused_blocks_kernel" $(find $d/s -name "*counter_collection.csv" | head -1) > $out/${tag}_${n}_pmc_sq.csv
  python3 $R/tools/pmc_traffic.py ${wl}@${m} $out/${tag}_${n}_pmc_fetch.csv $out/${tag}_${n}_pmc_write.csv $out/pmc_traffic.json > /dev/null
  echo "$n done: $(tail -1 $out/${tag}_${n}_bench_under_rocprofv3.json | cut -c1-200)"
 done
done
# the exact default bench command (all modes + CPU baseline leg), kernel trace only
d=/tmp/prof_default; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python3 $R/bench.py > $out/${tag}_bench_default_under_rocprofv3.json 2> $out/default_ks.err
cp $(find $d -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_default_kernel_stats.csv
echo "default bench done: $(tail -1 $out/${tag}_bench_default_under_rocprofv3.json | cut -c1-160)"
