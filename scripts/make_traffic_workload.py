#!/usr/bin/env python3
"""Per-kernel counter summary of one scripts/pmc_workload.sh run and its HBM traffic under bench.py's kernel labels.

    python scripts/make_traffic_workload.py gpurun_out/pmc_gat_after r06 products-khop3-gat5

Writes profiles/<tag>_pmc_<workload>.csv: one row per kernel and per launch-size CLUSTER (launches of one kernel that differ in
shape -- the dense layers and the row-sparse top layers of a GAT step run the same kernels on 294 k and on ~30 k rows -- are told
apart by their WAVE_CYCLES / FETCH value: sorted per-dispatch values are cut where they jump by more than 40 %), with the mean of
every counter per launch, and updates profiles/traffic.json[workload][label] = read + write bytes per launch of the LARGEST
cluster (read = 2 * FETCH_SIZE * 1024, write = WRITE_SIZE * 1024: MI355X_MICROARCH.md / profiles/r05_pmc_calibration.md)."""
import collections
import csv
import glob
import json
import os
import re
import sys

out, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
FILT = "shadow::"
# bench.py label -> substring of the kernel name (largest launch cluster)
LABELS = {
    "products-khop3-gat5": {
        "gat_fwd_tail_F256_H4": "gat_row_fwd_kernel<64 true", "gat_bwd_F256_H4": "gat_col_bwd_w8_kernel<64 false",
        "gat_bwd_map_F256_H4": "gat_col_bwd_w8_kernel<64 true",
        "gemm_nt2_gat_f16_N256": "gemm_nt_fused_kernel<8 5 2 1 false>", "gemm_nt2_gat_f16_N256_Ktail": "gemm_nt_fused_kernel<8 5 2 1 true>",
        "act_norm_bwd_nb2_F256": "act_norm_kernel<64 16 true 2>", "gemm_nt_f16_N256": "gemm_nt_fused_kernel<8 2 1 1 false>",
        "gemm_tn_f16_pair_N256": "gemm_tn_f16_kernel", "gemm_tn_split_N256_K128": "gemm_tn_coop_kernel<2 false>",
    },
    "products-ppr-sage5": {
        "gemm_act_norm_fwd_nb2_N256": "gemm_nt_fused_kernel<8 0 2 2 false>", "act_norm_bwd_nb2_F256": "act_norm_kernel<64 64 true 2>",
        "spmm_F256": "spmm_pipe_kernel<4 16 64>", "gemm_an_bwd_nb2_N256": "gemm_nt_fused_kernel<8 -1 1 2 false>",
        "gemm_tn_f16_pair_N256": "gemm_tn_f16_kernel", "segment_pool_F256": "segment_pool_fwd_kernel",
    },
    "arxiv-khop-sage5": {
        "gemm_an_bwd_nb2_N256": "gemm_nt_fused_kernel<8 1 1 2 false>", "gemm_act_norm_fwd_nb2_N256": "gemm_nt_fused_kernel<8 0 2 2 false>",
        "spmm_F256": "spmm_blockdiag_kernel<0>", "gemm_tn_f16_pair_N256": "gemm_tn_f16_kernel",
    },
}


def norm(kn):
    kn = kn[kn.index(FILT):] if FILT in kn else kn
    kn = re.sub(r"\(anonymous namespace\)::", "", kn)
    return re.sub(r"\(.*", "", kn)[:90].replace(", ", " ").replace(",", " ")


per = collections.defaultdict(lambda: collections.defaultdict(list))       # kernel -> counter -> per-dispatch values (dispatch order)
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if FILT not in r["Kernel_Name"]:
            continue
        per[norm(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))


def cut(vals):
    """indices of the clusters of the sorted values (ascending), cut at jumps > 40 %"""
    order = sorted(range(len(vals)), key=lambda i: vals[i])
    groups, cur = [], [order[0]]
    for a, b in zip(order, order[1:]):
        if vals[b] > 1.4 * vals[a] and vals[b] - vals[a] > 1024:
            groups.append(cur); cur = [b]
        else:
            cur.append(b)
    groups.append(cur)
    return groups


rows, traffic = [], {}
counters = sorted({c for k in per for c in per[k]})
for k in sorted(per):
    key = "FETCH_SIZE" if "FETCH_SIZE" in per[k] else counters[0]
    n = len(per[k][key])
    groups = cut(per[k][key]) if n > 1 else [[0]]
    for gi, g in enumerate(groups):
        frac = [i / n for i in g]                   # the same dispatches in the other passes: by rank position in dispatch order
        row = {"kernel": k, "cluster": f"{gi + 1}/{len(groups)}", "launches": len(g)}
        for c in counters:
            v = per[k].get(c, [])
            if len(v) == n:
                row[c] = round(sum(v[i] for i in g) / len(g))
            elif v:                                   # a pass that saw a different number of dispatches: whole-kernel mean
                row[c] = round(sum(v) / len(v))
        rows.append(row)
    big = groups[-1]
    if "FETCH_SIZE" in per[k] and "WRITE_SIZE" in per[k] and len(per[k]["WRITE_SIZE"]) == n:
        rd = 2 * 1024 * sum(per[k]["FETCH_SIZE"][i] for i in big) / len(big)
        wr = 1024 * sum(per[k]["WRITE_SIZE"][i] for i in big) / len(big)
        for label, sub in LABELS.get(workload, {}).items():
            if sub in k:
                traffic[label] = int(rd + wr)

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", f"{tag}_pmc_{workload}.csv")
with open(dst, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=["kernel", "cluster", "launches"] + counters)
    w.writeheader()
    for r in rows:
        w.writerow(r)
print("wrote", dst, len(rows), "rows")
path = os.path.join(root, "profiles", "traffic.json")
cur = json.load(open(path)) if os.path.exists(path) else {}
if traffic:
    cur[workload] = traffic
    cur.setdefault("_sources", {})[workload] = (f"profiles/{tag}_pmc_{workload}.csv (scripts/pmc_workload.sh: separate rocprofv3 --pmc passes of "
                                               f"bench.py --workload {workload} --steps 6 --warmup 2; largest launch cluster per kernel)")
    json.dump(cur, open(path, "w"), indent=1)
print(json.dumps(traffic, indent=1))
