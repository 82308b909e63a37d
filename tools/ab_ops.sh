#!/bin/bash
# A/B tagged experiment builds on the per-operator bench: tools/ab_ops.sh "tag1 tag2" <grep pattern>
tags=$1; pat=${2:-volume_cost}
for rep in 1 2; do
 for t in base $tags; do
  lib=zest-nerf_amd/libzest_hip_$t.so; [ "$t" = base ] && lib=zest-nerf_amd/libzest_hip.so
  echo "$t $(ZEST_HIP_LIB=$PWD/$lib python tools/bench_ops.py 2>/dev/null | grep "$pat" | cut -c1-160)"
 done
done
