#!/bin/bash
# kernel statistics of the whole-image loop (tools/bench_image.py) for profiles/
R=${GRAFT_REPO_ROOT:-$PWD}; out=$R/gpurun_out/image_prof; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ip
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ip -o ks -- python3 $R/tools/bench_image.py --precision 16 --images 3 > $out/image.json 2>/dev/null
cp $(find /tmp/ip -name "*kernel_stats.csv" | head -1) $out/r03_image16_kernel_stats.csv
head -24 $out/r03_image16_kernel_stats.csv | cut -c1-170
