#!/bin/bash
# kernel statistics of the fp32 training step (rocBLAS sgemm + HIP elementwise) for profiles/
R=${GRAFT_REPO_ROOT:-$PWD}; out=$R/gpurun_out/train_prof; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tp32
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp32 -o ks -- python3 $R/tools/bench_train.py --precision 32 --cpu-rays 0 --steps 3 > /dev/null 2>&1
cp $(find /tmp/tp32 -name "*kernel_stats.csv" | head -1) $out/r03_train32_step_kernel_stats.csv
head -30 $out/r03_train32_step_kernel_stats.csv | cut -c1-150
