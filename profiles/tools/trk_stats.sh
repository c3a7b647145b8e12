#!/bin/bash
# rocprofv3 --kernel-trace --stats of the native tracking / mapping iterations -> gpurun_out/prof_trk/kernel_stats.csv
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_trk
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 profiles/tracking_profile.py 300000 > "$out/stats.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(out + "/kernel_stats.csv", "w") as fh:
    fh.write("kernel,calls,avg_us,total_ms,percent\n")
    for r in rows:
        n = r["Name"].replace("void ", "").replace("mgs::", "").split("(")[0]
        fh.write(f"{n},{r['Calls']},{float(r['AverageNs']) / 1e3:.2f},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['Percentage']):.2f}\n")
print(open(out + "/kernel_stats.csv").read()[:1800])
PY
