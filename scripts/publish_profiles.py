#!/usr/bin/env python3
"""Copy the judged artefacts of scripts/collect_profiles.sh from gpurun_out/prof_<tag>/ into profiles/ (tracked):
    <tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats, as written
    <tag>_pmc_FETCH_SIZE.csv / _WRITE_SIZE.csv   per kernel: launches, mean counter value (KiB) per launch
    <tag>_rocprofv3_summary.md       the table of scripts/summarize_profiles.py
usage: scripts/publish_profiles.py gpurun_out/prof_r02 r02"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
ks = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == cname:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(os.path.join(dst, f"{tag}_pmc_{cname}.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Counter_Name", "launches", f"mean_{cname}_KiB_per_launch"])
        for k, (v, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            w.writerow([k, cname, c, round(v / c, 1)])
shutil.copy(os.path.join(out, f"summary_{tag}.md"), os.path.join(dst, f"{tag}_rocprofv3_summary.md"))
print("published", tag)
