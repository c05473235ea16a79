#!/usr/bin/env python3
"""Section 3 of docs/measurements/<tag>.md -- one block per BASELINE configuration -- from profiles/<tag>_bench_*.json.

    python scripts/workload_tables.py r06 [--write]      # --write replaces the block between the GENERATED markers of the file"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
ORDER = ["default", "arxiv-khop-sage5", "products-ppr-sage5", "products-khop3-gat5", "papers100M-ppr-sage5", "arxiv-khop-gcn3",
         "products-khop-sage5_b128"]
TITLE = {"default": "products-khop-sage5", "products-khop-sage5_b128": "products-khop-sage5 (128 roots)"}
out = []
for w in ORDER:
    p = os.path.join(ROOT, "profiles", f"{tag}_bench_{w}.json")
    if not os.path.exists(p):
        continue
    raw = open(p).read().strip()
    try:
        d = json.loads(raw)                                   # (pretty-printed by scripts/run_campaign.sh)
    except json.JSONDecodeError:
        d = json.loads(raw.splitlines()[-1])                  # (bench.py's own output: the line is the last one)
    rs, sa = d["roofline_step"], d.get("sampler_alone") or {}
    steps = d.get("instrumented_steps") or 1
    out.append(f"### {TITLE.get(w, w)} — {d['ms_per_step']:.2f} ms/step, {d['value'] / 1e6:.1f} M sampled nodes/s\n")
    out.append(f"host busy {d['host_busy_ms_per_step']:.2f} ms, kernels {rs['kernel_ms_per_step']:.2f} ms, whole-step fraction {rs['frac']:.3f}, "
               f"{d['config']['nodes_per_step']:.0f} nodes / step; sampler alone {sa.get('avg_ms')} ms per call ({sa.get('frac')} of HBM).\n")
    out.append("| kernel class | launches / step | µs | alg. frac of 8 TB/s |\n|---|---|---|---|")
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ms"])[:10]:
        out.append(f"| `{k}` | {v['launches'] / steps:.2f} | {v['avg_ms'] * 1e3:.1f} | {v['frac']} |")
    out.append("")
text = "\n".join(out)
if "--write" in sys.argv:
    path = os.path.join(ROOT, "docs", "measurements", f"{tag}.md")
    s = open(path).read()
    b, e = "<!-- BEGIN GENERATED scripts/workload_tables.py -->", "<!-- END GENERATED workload_tables -->"
    i, j = s.index(b) + len(b), s.index(e)
    open(path, "w").write(s[:i] + "\n" + text + "\n" + s[j:])
    print("updated", path)
else:
    print(text)
