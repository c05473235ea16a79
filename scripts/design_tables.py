#!/usr/bin/env python3
"""The measured tables of DESIGN.md, generated from the committed artefacts under profiles/ (nothing hand-copied):

    python scripts/design_tables.py r03            # prints markdown
    python scripts/design_tables.py r03 --write    # replaces the block between the GENERATED markers in DESIGN.md

Sources: profiles/<tag>_bench_default.json (the bench line: HIP-event times per kernel class on its launch stream, the
algorithmic bytes / flops of SURVEY.md 8(d)), profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the
same command), profiles/traffic.json (HBM bytes per launch from the separate FETCH_SIZE / WRITE_SIZE passes),
profiles/<tag>_bench_<workload>.json (the other BASELINE configurations)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"

# bench.py kernel class -> substring of the rocprofv3 kernel name
ROC = {
    "gemm_an_bwd_nb2_N256": "gemm_nt_fused_kernel<8, 1, 1, 2, false>",
    "gemm_act_norm_fwd_nb2_N256": "gemm_nt_fused_kernel<8, 0, 2, 2, false>",
    "gemm_act_norm_fwd_nb2_N256_Ktail": "gemm_nt_fused_kernel<8, 0, 2, 2, true>",
    "gemm_tn_split_N256": "gemm_tn_coop_kernel<4, true>",
    "gemm_tn_f16_N256": "gemm_tn_f16_kernel",
    "gemm_tn_f16_pair_N256": "gemm_tn_f16_kernel",
    "gemm_tn_split_N256_K128": "gemm_tn_split_kernel<2>",
    "act_norm_bwd_nb2_F256": "act_norm_kernel<64, 64, true, 2>",
    "gather_F100": "gather_rows_drop_kernel<32>",
    "spmm_F256": "spmm_blockdiag_kernel<false>",
    "spmm_F100": "spmm_blockdiag_kernel<false>",
    "spmm_rows_F256": "spmm_pipe_kernel",
}
HBM, MFMA6, MFMA3 = 8000.0, 2500.0 / 6.0, 2500.0 / 3.0


def mfma_roof(kernel):
    """fp32-equivalent roof of a GEMM class: six bf16 products per multiply-add (tn, plain nt) or three fp16 ones."""
    return MFMA3 if kernel.startswith(("gemm_act_norm", "gemm_an_bwd", "gemm_nt_f16", "gemm_tn_f16")) else MFMA6


def main():
    d = json.load(open(os.path.join(P, f"{tag}_bench_default.json")))
    roc = {}
    for r in csv.DictReader(open(os.path.join(P, f"{tag}_kernel_stats.csv"))):
        roc[r["Name"]] = float(r["AverageNs"]) / 1e3
    traffic = json.load(open(os.path.join(P, "traffic.json"))).get("products-khop-sage5", {})
    out = []
    out.append(f"`bench.py` (no flags: products-shape k-hop d2 b20 + SAGE-5 dim 256, 1 024 roots): **{d['ms_per_step']} ms/step = "
               f"{d['train_steps_per_sec']} steps/s = {d['value'] / 1e6:.1f} M sampled nodes/s** through the full train step "
               f"({d['config']['nodes_per_step']:.0f} nodes, {d['config']['edges_per_step']:.0f} edges per step); host busy "
               f"{d['host_busy_ms_per_step']} ms of it (enqueue {d['host_enqueue_ms_per_step']} ms incl. the blocked count read-back).")
    out.append("")
    out.append("| kernel class (bench line) | launches / step | avg us, HIP events | avg us, rocprofv3 | algorithmic MB | HBM MB (PMC) | PMC / alg. | alg. GB/s | frac of 8 TB/s | TFLOP/s fp32-equiv. | frac of the split's roof (2500/3 fp16 pieces, 2500/6 bf16) |")
    out.append("|---|---|---|---|---|---|---|---|---|---|---|")
    K = d["kernels"]
    def prof_steps(line):
        """instrumented steps behind the line's kernel statistics"""
        if line.get("instrumented_steps"):
            return int(line["instrumented_steps"])
        mo = re.match(r"(\d+) instrumented", (line.get("roofline_north_star") or line.get("roofline") or {}).get("measured_over", ""))
        return int(mo.group(1)) if mo else (10 if line["steps"] >= 10 else line["steps"])
    steps_prof = prof_steps(d)
    for k, v in sorted(K.items(), key=lambda kv: -kv[1]["total_ms"]):
        if v["total_ms"] < 0.5 and not k.startswith(("sg_", "gather")):
            continue
        rn = ROC.get(k)
        rocus = next((f"{t:.1f}" for n, t in roc.items() if rn and rn in n), "—") if k not in ("spmm_F100", "spmm_F256") else (next((f"{t:.1f} (both widths)" for n, t in roc.items() if ("spmm_blockdiag_kernel<false>" in n or "spmm_blockdiag_kernel<0>" in n)), "—") if k == "spmm_F256" else "—")
        by = None
        for name, e in (("roofline_hbm", d.get("roofline_hbm")), ("roofline_mfma", d.get("roofline_mfma"))):
            if e and e.get("kernel") == k:
                by = e.get("bytes_per_launch")
        alg_mb = v["alg_GBps"] * v["avg_ms"] * 1e-3 * 1e3          # GB/s * s = GB -> MB
        tr = traffic.get(k)
        tf = v.get("alg_TFLOPs")
        trs = f"{tr / 1e6:.0f} | {tr / 1e6 / alg_mb:.2f}" if (tr and alg_mb > 0) else "— | —"
        out.append(f"| `{k}` | {v['launches'] / steps_prof:.1f} | {v['avg_ms'] * 1e3:.1f} | {rocus} | {alg_mb:.0f} | {trs}"
                   + f" | {v['alg_GBps']:.0f} | {v['frac']:.3f} | " + (f"{tf:.1f} | {tf / mfma_roof(k):.3f} |" if tf else "— | — |"))
    r = d.get("roofline_north_star") or d["roofline"]
    dom, rs = d["roofline"], d.get("roofline_step")
    out.append("")
    if d.get("roofline_north_star"):
        out.append(f"`roofline` of the line = the dominant kernel class by total time: `{dom['kernel']}`, {dom.get('launches_per_step')} launches per step x "
                   f"{dom['avg_ms'] * 1e3:.1f} us = {dom.get('share_of_kernel_time', 0) * 100:.0f} % of the kernel time; {dom['achieved']:.0f} GB/s algorithmic = "
                   f"**{dom['frac']:.3f} of 8 TB/s** (arithmetic intensity {dom.get('arith_intensity_flop_per_byte')} flop/B against a ridge of "
                   f"{dom.get('ridge_flop_per_byte')}: HBM is its roof; {dom.get('mfma_frac')} of the fp16 split's matrix-core roof beside it).")
        if rs:
            out.append(f"`roofline_step`: every kernel class's own algorithmic bytes, {rs['bytes_per_step'] / 1e9:.2f} GB per step, over the timed "
                       f"{rs['ms_per_step']} ms = {rs['achieved']:.0f} GB/s = **{rs['frac']:.3f} of 8 TB/s** ({rs['kernel_ms_per_step']} ms of it inside the hand-written kernels).")
    out.append(f"North-star aggregate (k-hop sample + feature gather + SAGE aggregates, `roofline_north_star` of the bench line): "
               f"{r['bytes_per_step'] / 1e6:.0f} MB / {r['ms_per_step']} ms = {r['achieved']:.0f} GB/s = **{r['frac']:.3f} of 8 TB/s**; "
               + (f"PMC traffic of those kernels {r['traffic'] / 1e6:.0f} MB per step." if r.get("traffic") else ""))
    sa = d.get("sampler_alone")
    if sa:
        out.append(f"Sampler kernels with the GPU to themselves: {sa['avg_ms']} ms per call of {sa.get('batches_per_call', 1)} x 1 024 roots "
                   f"({sa.get('us_per_subgraph')} us per subgraph) = {sa['alg_GBps']:.0f} GB/s = {sa['frac']} of peak (+ relocation {sa['relocate_avg_ms']} ms).")
    cb = d.get("cpu_baseline") or {}
    if cb.get("value"):
        out.append(f"CPU baseline (the reference's own C++/OpenMP sampler, `oracle/_ref`, best of a thread sweep): {cb['value'] / 1e6:.2f} M sampled "
                   f"nodes/s on {cb.get('cores')} threads; one thread {float(cb.get('one_thread') or 0) / 1e6:.2f} M (sweep {cb.get('sweep')}).")
    cs = d.get("cpu_baseline_train_step") or {}
    if cs.get("value"):
        out.append(f"CPU train step (`oracle/cpu_train_step.py`, kind port, {'extrapolated from ' + str(cs.get('measured_fraction_of_batch')) + ' of a batch' if cs.get('extrapolated') else 'whole batch'}): "
                   f"{cs['value']} steps/s on {cs.get('cores')} threads.")
    dt = d.get("dense_top_backward") or {}
    if dt.get("ms_per_step"):
        out.append(f"The same step with the top layer's backward pass on every row (`--dense-top-backward`, 10 steps after the timed region): {dt['ms_per_step']} ms/step.")
    tl = d.get("target_only_tail") or {}
    if tl.get("ms_per_step"):
        out.append(f"Opt-in target-only tail (never part of `value`): {tl['ms_per_step']} ms/step.")
    out.append("")
    out.append("Other BASELINE configurations (`bench.py --workload ...`, `profiles/" + tag + "_bench_<workload>.json`; rocprofv3 kernel "
               "statistics of the PPR, GAT and arxiv SAGE-5 runs: `profiles/" + tag + "_kernel_stats_<workload>.csv`, "
               "`scripts/collect_workload_stats.sh`):")
    out.append("")
    out.append("| workload | ms / step | sampled nodes / s | host busy ms | north-star frac | three longest kernel classes (avg us x launches per step, frac) |")
    out.append("|---|---|---|---|---|---|")
    for f in sorted(os.listdir(P)):
        m = re.match(rf"{tag}_bench_(.+)\.json", f)
        if not m or m.group(1) == "default":
            continue
        w = json.load(open(os.path.join(P, f)))
        top = sorted(w["kernels"].items(), key=lambda kv: -kv[1]["total_ms"])[:3]
        sp = prof_steps(w)
        tops = "; ".join(f"`{k}` {v['avg_ms'] * 1e3:.0f} x {v['launches'] / sp:.0f} ({v['frac']:.2f})" for k, v in top)
        out.append(f"| {m.group(1)} | {w['ms_per_step']} | {w['value'] / 1e6:.2f} M | {w.get('host_busy_ms_per_step')} | {(w.get('roofline_north_star') or w['roofline'])['frac']} | {tops} |")
    text = "\n".join(out)
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        b, e = "<!-- BEGIN GENERATED scripts/design_tables.py -->", "<!-- END GENERATED -->"
        i, j = s.index(b) + len(b), s.index(e)
        open(path, "w").write(s[:i] + "\n" + text + "\n" + s[j:])
        print("DESIGN.md updated")
    else:
        print(text)


if __name__ == "__main__":
    main()
