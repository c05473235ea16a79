#!/usr/bin/env python3
"""PMC calibration report: FETCH_SIZE / WRITE_SIZE of kernels with KNOWN byte counts.

    scripts/micro/calib_report.py <dir with calib_stdout.jsonl, pmc_fetch/, pmc_write/>  >  profiles/r05_pmc_calibration.md

`stream_ceiling calib` prints one JSON line per kernel with the bytes it moves (algorithmic, and -- for gathers -- the bytes
of the 64-B / 128-B lines its rows touch); the two rocprofv3 --pmc passes give the counter per kernel name.  The table
shows counter x 1024 next to those byte counts and the factor that maps the counter onto each."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
known = [json.loads(l) for l in open(os.path.join(d, "calib_stdout.jsonl")) if l.startswith("{")]


def counters(sub, cname):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == cname:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")


def lookup(acc, name):
    ks = [k for k in acc if name.replace(" ", "") in k.replace(" ", "")]
    if not ks:
        return None
    v = acc[ks[0]]
    return sum(v) / len(v)


print("# PMC calibration on kernels with known byte counts (`scripts/micro/stream_ceiling.hip calib`, one launch each)\n")
print("Counters in KiB (rocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separate passes).  `x` = bytes / (counter x 1024): the factor a PMC reading "
      "has to be multiplied with to give that byte count.\n")
print("| kernel | access shape | FETCH_SIZE KiB | algorithmic read MB (x) | 64-B lines touched MB (x) | 128-B lines touched MB (x) | WRITE_SIZE KiB | algorithmic write MB (x) | 64-B / 128-B lines written MB (x) |")
print("|---|---|---|---|---|---|---|---|---|")
for k in known:
    name = k["calib"]
    f, w = lookup(fetch, name), lookup(write, name)

    def cell(b, c):
        if b is None or not b:
            return "—"
        if c is None or c == 0:
            return f"{b / 1e6:.1f} (no counter)"
        return f"{b / 1e6:.1f} ({b / (c * 1024):.2f})"
    rb = k.get("read_bytes", 0) + k.get("index_bytes", 0)
    r64 = k.get("read_bytes_64B_lines")
    r128 = k.get("read_bytes_128B_lines")
    if r64:
        r64 += k.get("index_bytes", 0)
        r128 += k.get("index_bytes", 0)
    wl = "—"
    if k.get("write_bytes_64B_lines"):
        wl = f"{cell(k['write_bytes_64B_lines'], w)} / {cell(k['write_bytes_128B_lines'], w)}"
    print(f"| `{name}` | {k.get('shape', 'wide coalesced stream (16 B / lane)')} | {f if f is not None else '—'} | {cell(rb, f)} | {cell(r64, f)} | {cell(r128, f)} | "
          f"{w if w is not None else '—'} | {cell(k.get('write_bytes', 0), w)} | {wl} |")
