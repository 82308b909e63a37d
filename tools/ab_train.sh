#!/bin/bash
# A/B the tagged experiment builds of libzest_hip on the training kernels (per-kernel times and the whole step).
# usage: tools/ab_train.sh "tag1 tag2 ..."
for t in base $1 base $1; do
  lib=zest-nerf_amd/libzest_hip_$t.so; [ "$t" = base ] && lib=zest-nerf_amd/libzest_hip.so
  echo "== $t"
  ZEST_HIP_LIB=$PWD/$lib python tools/time_train16.py 2>/dev/null | tail -1
  ZEST_HIP_LIB=$PWD/$lib python tools/time_train16.py mlp_static_sf_mvs40 2>/dev/null | tail -1
  ZEST_HIP_LIB=$PWD/$lib python tools/bench_train.py --precision 16 --cpu-rays 0 --steps 20 2>/dev/null | tail -1 | cut -c60-150
done
