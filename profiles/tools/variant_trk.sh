#!/bin/bash
# profiles/tracking_profile.py (300 k map) with every library variant under scratch/variants: second-order line only
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so)
  MGS_LIB_PATH=$PWD/$f python3 profiles/tracking_profile.py 300000 > gpurun_out/trk_$n.txt 2>&1
  echo "== $n"; grep "second order\|first order" gpurun_out/trk_$n.txt
done
