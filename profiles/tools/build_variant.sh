#!/bin/bash
# usage: scratch/build_variant.sh <name> <extra hipcc flags...>   -> scratch/variants/lib_<name>.so
set -e
name=$1; shift
cd /root/repo/monogs_amd/csrc
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -shared -fPIC "$@" \
  raster_forward.hip raster_backward.hip raster_cabi.hip knn.hip tracking.hip map_update.hip \
  -o /root/repo/scratch/variants/lib_$name.so
