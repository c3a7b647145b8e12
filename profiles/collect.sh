#!/bin/bash
# Collects the profiles quoted in DESIGN.md on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag>
# 1. rocprofv3 --kernel-trace --stats           -> gpurun_out/prof_<tag>/stats  (per-kernel durations)
# 2. rocprofv3 --kernel-trace --pmc FETCH_SIZE  -> gpurun_out/prof_<tag>/fetch  (separate pass)
# 3. rocprofv3 --kernel-trace --pmc WRITE_SIZE  -> gpurun_out/prof_<tag>/write  (separate pass)
# and summarises them (profiles/summarise.py) into gpurun_out/prof_<tag>/*.{csv,txt,json}.
set -e
tag=${1:-run}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
BENCH="python3 bench.py --steps 30 --warmup 5 --lean --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- $BENCH > "$out/stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$out/fetch" -o run -- $BENCH > "$out/fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$out/write" -o run -- $BENCH > "$out/write.log" 2>&1
python3 profiles/summarise.py "$out"
# 4. SQ counters of the blend kernels (VALU / SALU / LDS instruction counts, busy and wait cycles)
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$out/sq_$name" -o run -- $BENCH > "$out/sq_$name.log" 2>&1 || echo "pmc group failed: $grp"
done
python3 profiles/summarise.py "$out" sq
