#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy figures of the product kernels, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).  usage: python profiles/tools/resource_usage.py [file.hip ...] [-D...]
Prints: kernel, VGPRs, AGPRs, SGPRs, scratch [B/lane], occupancy [waves/SIMD], LDS [B/block]."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "monogs_amd", "csrc")
files = [a for a in sys.argv[1:] if a.endswith(".hip")] or ["raster_forward.hip", "raster_backward.hip", "tracking.hip",
                                                           "map_update.hip", "knn.hip"]
flags = [a for a in sys.argv[1:] if not a.endswith(".hip")]
KEYS = (("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("SGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"),
        ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "lds"))
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-c", f,
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *flags], cwd=CSRC,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    name, row = None, {}

    def flush():
        if name:
            print(f"{name[:64]:64s} " + " ".join(f"{short} {row.get(k, '?'):>4}" for k, short in KEYS))
    for line in r.stdout.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            flush()
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True,
                                  text=True).stdout.strip() or m.group(1)
            name = re.sub(r"\(.*$", "", re.sub(r"^void mgs::", "", name))
            row = {}
            continue
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+([A-Za-z][^:]*): (\d+)", line) or re.search(r"remark:\s+.*?([A-Z][A-Za-z \[\]/]+): (\d+)", line)
        if m:
            row[m.group(1).strip()] = m.group(2)
    flush()
    if r.returncode != 0:
        print(r.stdout[-3000:])
