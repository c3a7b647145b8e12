#!/bin/bash
# runs bench.py --lean with every library variant under scratch/variants and prints the blend_bwd time
cd $GRAFT_REPO_ROOT
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so)
  MGS_LIB_PATH=$PWD/$f python bench.py --lean --steps 50 --warmup 10 > gpurun_out/r03_var_$n.json 2> gpurun_out/r03_var_$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r03_var_$n.json") if l.startswith("{")][-1])
    print("$n", "fps", d["value"], "kernels", d["kernels_us"])
except Exception as e:
    print("$n", "FAILED", e)
PY
done
