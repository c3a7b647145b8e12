#!/bin/bash
# A/B of library variants under scratch/variants (profiles/tools/build_variant.sh): for each one
#   * bench.py --lean --steps 50 (fps, kernels_us from the in-library HIP events)
#   * rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) -> corrected HBM bytes per launch
#     of the blend kernels, (2 FETCH + WRITE) * 1024 (MI355X_MICROARCH.md, HBM section)
# usage (through gpurun, from the repo root): profiles/tools/variant_pmc.sh [tag]   -> gpurun_out/<tag>_variants.txt
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$root"
out=gpurun_out/${tag}_variants.txt
: > $out
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so)
  export MGS_LIB_PATH=$PWD/$f
  python3 bench.py --lean --steps 50 --warmup 10 > gpurun_out/${tag}_var_$n.json 2> gpurun_out/${tag}_var_$n.err
  d=gpurun_out/${tag}_pmc_$n
  rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $d/fetch -o run -- python3 bench.py --steps 20 --warmup 5 --lean --profile-steps 0 > $d/fetch.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $d/write -o run -- python3 bench.py --steps 20 --warmup 5 --lean --profile-steps 0 > $d/write.log 2>&1
  python3 - "$n" "$d" "gpurun_out/${tag}_var_$n.json" >> $out <<'PY'
import collections, csv, glob, json, sys
n, d, j = sys.argv[1:4]
try:
    r = json.loads([l for l in open(j) if l.startswith("{")][-1])
    head = f"{n}: fps {r['value']} kernels_us {r['kernels_us']}"
except Exception as e:
    head = f"{n}: bench FAILED {e}"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == ctr:
                k = row["Kernel_Name"].replace("void ", "").replace("mgs::", "").split("(")[0]
                agg[k][ctr].append(float(row["Counter_Value"]))
print(head)
for k, v in sorted(agg.items()):
    if not k.startswith("k_blend"):
        continue
    fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
    wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    print(f"    {k}: launches {len(v['FETCH_SIZE'])} FETCH {fe:.0f} KB WRITE {wr:.0f} KB -> corrected {(2 * fe + wr) * 1024 / 1e6:.1f} MB, uncorrected {(fe + wr) * 1024 / 1e6:.1f} MB")
PY
done
cat $out
