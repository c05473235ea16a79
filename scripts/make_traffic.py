#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of scripts/collect_profiles.sh: HBM bytes per launch (read + write) under
the kernel labels bench.py uses.

    python scripts/make_traffic.py gpurun_out/prof_r02 r02 products-khop-sage5

Read bytes = 2 * FETCH_SIZE * 1024, write bytes = WRITE_SIZE * 1024 (MI355X_MICROARCH.md: the counters are in KiB and
on gfx950 FETCH_SIZE counts a 128-byte request of a wide coalesced read as 64 bytes).  rocprofv3 only knows kernel
names; launches of one kernel that differ in shape (the block-diagonal SpMM at F = 100 / F = 256, the nt GEMM at
K = 256 / K = 512) are told apart by their counter value: the sorted per-dispatch values are cut where they jump by
more than 15 %."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]


def per_dispatch(sub, cname):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == cname and "shadow::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def clusters(vals):
    """[(mean, count)] of the value groups, ascending."""
    v = sorted(vals)
    groups, cur = [], [v[0]]
    for x in v[1:]:
        if x > 1.15 * cur[-1] and x - cur[-1] > 1024:
            groups.append(cur); cur = [x]
        else:
            cur.append(x)
    groups.append(cur)
    return [(sum(g) / len(g), len(g)) for g in groups]


fetch = per_dispatch("pmc_fetch", "FETCH_SIZE")
write = per_dispatch("pmc_write", "WRITE_SIZE")


def find(sub):
    ks = [k for k in fetch if sub in k]
    assert len(ks) == 1, (sub, ks)
    return ks[0]


def rw(sub, which=None, of=None):
    """read+write bytes per launch of the kernel whose name contains `sub`; `which`/`of`: cluster index / expected
    cluster count when its launches mix shapes (None = mean over all launches)."""
    k = find(sub)
    res = []
    for acc, scale in ((fetch, 2 * 1024.0), (write, 1024.0)):
        cl = clusters(acc[k])
        if which is None:
            res.append(scale * sum(acc[k]) / len(acc[k]))
        else:
            if len(cl) != of:
                # cut at the (of - 1) largest relative jumps of the sorted values instead
                v = sorted(acc[k])
                jumps = sorted(range(1, len(v)), key=lambda i: -(v[i] / max(v[i - 1], 1.0)))[:of - 1]
                cuts = [0] + sorted(jumps) + [len(v)]
                cl = [(sum(v[a:b]) / (b - a), b - a) for a, b in zip(cuts[:-1], cuts[1:])]
            res.append(scale * cl[which][0])
    return res


t = {}
notes = {}


def put(label, sub, which=None, of=None):
    """t[label] = read + write bytes per launch of the kernel whose name contains `sub` (a tuple: the first alternative the
    run has -- kernel names change with their template arguments); skipped when the run has none."""
    for one in ((sub,) if isinstance(sub, str) else sub):
        if [k for k in fetch if one in k]:
            t[label] = sum(rw(one, which, of))
            return


sel, plan, scan = rw("sg_select_lds_kernel"), rw("sg_plan_kernel"), rw("sg_scan_plain_kernel")
t["sg_sample_pipeline"] = sum(sel) + sum(plan) + sum(scan)
notes["sg_sample_pipeline"] = dict(select=sel, plan=plan, scan=scan)
put("sg_relocate_kernel", "sg_relocate_kernel")
put("gather_F100", "gather_rows_drop_kernel")
put("spmm_F100", ("spmm_blockdiag_kernel<0>", "spmm_blockdiag_kernel<false>"), 0, 2)
put("spmm_F256", ("spmm_blockdiag_kernel<0>", "spmm_blockdiag_kernel<false>"), 1, 2)
put("act_norm_fwd_nb2_F256", "act_norm_kernel<64, 64, false, 2>")
put("act_norm_bwd_nb2_F256", "act_norm_kernel<64, 64, true, 2>")
put("gemm_nt_split_N256", "gemm_nt_split_kernel<1, 8, 1, 4, false>")
put("gemm_nt_split_N256_Ktail", "gemm_nt_split_kernel<1, 8, 1, 4, true>")
# round 3: the GEMM-epilogue kernels (csrc/gemm_fused.hip) and the cooperative-split weight gradient
put("gemm_act_norm_fwd_nb2_N256", "gemm_nt_fused_kernel<8, 0, 2, 2, false>")
put("gemm_act_norm_fwd_nb2_N256_Ktail", "gemm_nt_fused_kernel<8, 0, 2, 2, true>")
# round 4: the same kernel runs the K = 2F product (two launches per step) and the K = F product with the sparse addend (one)
if [k for k in fetch if "gemm_nt_fused_kernel<8, 1, 1, 2, false>" in k] and len(clusters(fetch[find("gemm_nt_fused_kernel<8, 1, 1, 2, false>")])) >= 2:
    put("gemm_an_bwd_corr_nb2_N256", "gemm_nt_fused_kernel<8, 1, 1, 2, false>", 0, 2)
    put("gemm_an_bwd_nb2_N256", "gemm_nt_fused_kernel<8, 1, 1, 2, false>", 1, 2)
else:
    put("gemm_an_bwd_nb2_N256", "gemm_nt_fused_kernel<8, 1, 1, 2, false>")
put("spmm_rows_F256", ("spmm_pipe_kernel", "spmm_blockdiag_kernel<2>", "spmm_blockdiag_kernel<true>"))   # (round 5: the filtered transposed structure on the pipelined kernel)
if [k for k in fetch if "gemm_tn_f16_kernel" in k]:
    put("gemm_tn_f16_pair_N256", "gemm_tn_f16_kernel")
if [k for k in fetch if "gemm_tn_coop_kernel<4, true>" in k]:
    put("gemm_tn_split_N256", "gemm_tn_coop_kernel<4, true>")
else:
    put("gemm_tn_split_N256", "gemm_tn_split_kernel<4>")
put("gemm_tn_split_N256_K128", "gemm_tn_split_kernel<2>")
t = {k: int(v) for k, v in t.items()}

path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
cur = {}
if os.path.exists(path):
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
cur[workload] = t
cur["_source"] = (f"profiles/{tag}_pmc_FETCH_SIZE.csv + {tag}_pmc_WRITE_SIZE.csv (separate rocprofv3 --pmc passes of bench.py --steps 10 "
                  "--warmup 3, scripts/collect_profiles.sh); read bytes = 2*FETCH_SIZE*1024 (gfx950 correction, MI355X_MICROARCH.md; calibrated on known-byte kernels in every access shape of this repository, profiles/r05_pmc_calibration.md: 2*FETCH_SIZE*1024 = bytes of the 128-B lines touched, WRITE_SIZE*1024 = bytes written at 32-B granularity), "
                  "write bytes = WRITE_SIZE*1024; per-launch means by kernel; launches of one kernel that differ in shape (SpMM F = 100 / 256) "
                  "are separated by counter value (scripts/make_traffic.py); "
                  "sg_sample_pipeline = select + plan + scan kernels of one call")
json.dump(cur, open(path, "w"), indent=1)
print(json.dumps(t, indent=1))
print(json.dumps(notes, indent=1))
