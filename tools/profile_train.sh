#!/bin/bash
# training-step numbers and kernel stats for profiles/ (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}; out=$R/gpurun_out/train_prof; mkdir -p $out
python3 $R/tools/bench_train.py --precision 16 --cpu-rays 0 --steps 20 > $out/train16.json 2>/dev/null
python3 $R/tools/bench_train.py --precision 16 --cpu-rays 0 --steps 10 --frames 5 > $out/train16_5f.json 2>/dev/null
python3 $R/tools/bench_train.py --precision 32 --cpu-rays 0 --steps 10 > $out/train32.json 2>/dev/null
python3 $R/tools/time_train16.py > $out/train16_kernels.json 2>/dev/null
python3 $R/tools/time_train16.py mlp_static_sf_mvs40 >> $out/train16_kernels.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -o ks -- python3 $R/tools/bench_train.py --precision 16 --cpu-rays 0 --steps 5 > /dev/null 2>&1
cp $(find /tmp/tp -name "*kernel_stats.csv" | head -1) $out/r03_train16_step_kernel_stats.csv
tail -1 $out/train16.json; tail -1 $out/train16_5f.json; tail -1 $out/train32.json; cat $out/train16_kernels.json
