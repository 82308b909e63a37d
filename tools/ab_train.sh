#!/bin/bash
# A/B tagged experiment builds on the training-step bench: tools/ab_train.sh "tag1 tag2"
tags=$1
for rep in 1 2; do
 for t in base $tags; do
  lib=zest-nerf_amd/libzest_hip_$t.so; [ "$t" = base ] && lib=zest-nerf_amd/libzest_hip.so
  echo "$t $(ZEST_HIP_LIB=$PWD/$lib python tools/bench_train.py --precision 16 --cpu-rays 0 --steps 20 2>/dev/null | tail -1 | cut -c1-200)"
 done
done
