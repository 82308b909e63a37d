#!/bin/bash
# The measurement set a round's DESIGN.md / README quote, in one gpurun call (run from the repo root):
#   bash tools/final_round.sh r03
# rocprofv3 kernel stats + PMC passes per workload and operand type (tools/profile_round.sh), the whole-image loop,
# the training step (tools/profile_train.sh), the configs[3] shapes, in-kernel stamps (needs libzest_hip_stamps.so).
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}; out=$R/gpurun_out/final_$tag; mkdir -p $out
bash tools/profile_round.sh $tag "nsff_static_1024x128 nsff_static_mvs_1024x128 nsff_zest_val_1024x128" "bf16 f16x3" 2>&1 | tail -8
bash tools/profile_round.sh $tag "nsff_static_mvs_grid_1024x128 nsff_zest_val_grid_1024x128" "bf16" 2>&1 | tail -3
cd $R
python3 bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err; tail -1 $out/${tag}_bench_default.json | cut -c1-200
python3 bench.py --workload zest_val_4096x192 --steps 50 --warmup 5 --no-cpu-baseline --no-modes > $out/${tag}_bench_zest_val_4096x192.json 2>/dev/null
python3 bench.py --workload zest_val_4096x192 --rays 512 --steps 200 --warmup 20 --no-cpu-baseline --no-modes > $out/${tag}_bench_zest_val_4096x192_shard512.json 2>/dev/null
python3 bench.py --workload dtu_static_8192x128 --mode f16 --steps 50 --warmup 5 --no-cpu-baseline --no-modes > $out/${tag}_bench_dtu_static_8192x128_f16.json 2>/dev/null
for f in zest_val_4096x192 zest_val_4096x192_shard512 dtu_static_8192x128_f16; do tail -1 $out/${tag}_bench_$f.json | cut -c1-160; done
python3 tools/bench_image.py --precision 16 | tail -1 > $out/${tag}_image16.json; python3 tools/bench_image.py --precision 32 | tail -1 > $out/${tag}_image32.json; cat $out/${tag}_image16.json $out/${tag}_image32.json
bash tools/profile_train.sh 2>&1 | tail -5
if [ -f zest-nerf_amd/libzest_hip_stamps.so ]; then
  for wl in nsff_static_1024x128 nsff_static_mvs_1024x128 nsff_zest_val_1024x128; do
    ZEST_HIP_LIB=$PWD/zest-nerf_amd/libzest_hip_stamps.so python3 tools/stamps.py $wl 2>/dev/null | tail -7
  done > $out/${tag}_stamps.txt; cat $out/${tag}_stamps.txt
fi
python3 tools/bench_ops.py > $out/${tag}_ops.jsonl 2>/dev/null; cat $out/${tag}_ops.jsonl | cut -c1-160
# volume builders: the HIP regularisation net / feature pyramid (kernel stats, PMC), and the whole-generator training step
bash tools/prof_costreg.sh 2>&1 | tail -3
bash tools/prof_costreg_pmc.sh 1 2>&1 | tail -3
python3 tools/prof_builder.py --serial > $out/${tag}_builder_hip_costreg_kernels.txt 2>/dev/null; head -3 $out/${tag}_builder_hip_costreg_kernels.txt
(python3 tools/bench_generator_train.py --precision 16 | head -12; echo "--library-costreg"; python3 tools/bench_generator_train.py --precision 16 --library-costreg | head -8) > $out/${tag}_generator_train_step_hip_costreg.txt 2>/dev/null; head -3 $out/${tag}_generator_train_step_hip_costreg.txt
