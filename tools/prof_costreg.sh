#!/bin/bash
# rocprofv3 kernel statistics of the HIP regularisation net (tools/bench_costreg.py) and of the whole-image loop, for profiles/
R=${GRAFT_REPO_ROOT:-$PWD}; out=$R/gpurun_out/costreg_prof; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for p in 1 3; do
  rm -rf /tmp/cp$p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp$p -o ks -- python3 $R/tools/bench_costreg.py --passes $p > $out/costreg_p$p.txt 2>/dev/null
  cp $(find /tmp/cp$p -name "*kernel_stats.csv" | head -1) $out/r03_costreg_passes${p}_kernel_stats.csv
done
rm -rf /tmp/ip
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ip -o ks -- python3 $R/tools/bench_image.py --precision 16 --images 5 > $out/image.json 2>/dev/null
cp $(find /tmp/ip -name "*kernel_stats.csv" | head -1) $out/r03_image16_hip_builders_kernel_stats.csv
head -12 $out/r03_costreg_passes1_kernel_stats.csv | cut -c1-150
head -14 $out/r03_image16_hip_builders_kernel_stats.csv | cut -c1-150
tail -1 $out/image.json
