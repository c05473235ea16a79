#!/usr/bin/env bash
# Where does the GPU idle inside a step?  rocprofv3 --kernel-trace of bench.py, then per stream: busy time, idle time and the largest
# gaps (with the kernels on either side) over the timed region's steps.
#   usage: scripts/gap_analysis.sh <tag> [bench args...]
set -uo pipefail
TAG="${1:-gap}"; shift || true
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/gap_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-tail --no-other-workloads "$@" > "$OUT/bench.log" 2>&1
python - "$OUT" <<'PY' | tee "$OUT/gaps.txt"
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "0"))) for r in rows]
ev.sort()
# the timed region: find the steady part -- take the last 60 % of the clip_adam launches as step markers
adam = [e for e in ev if "clip_adam" in e[2]]
if len(adam) < 12:
    print("not enough steps"); sys.exit(0)
t0, t1 = adam[len(adam) // 3][1], adam[-3][1]
nsteps = len(adam) - 3 - len(adam) // 3
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print(f"window: {nsteps} steps, {(t1 - t0) / nsteps / 1e6:.3f} ms per step, {len(win) / nsteps:.1f} kernels per step")
byq = collections.defaultdict(list)
for e in win: byq[e[3]].append(e)
# union busy time over all queues
iv = sorted((e[0], e[1]) for e in win)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e, s)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU busy (any queue) {busy / nsteps / 1e6:.3f} ms per step, idle {(t1 - t0 - busy) / nsteps / 1e6:.3f} ms per step in {len(gaps) / nsteps:.1f} gaps per step")
for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print(f"queue {q}: {len(es) / nsteps:.1f} kernels per step, {sum(e[1] - e[0] for e in es) / nsteps / 1e6:.3f} ms per step")
# gap histogram by the kernel that FOLLOWS the gap
after = collections.defaultdict(lambda: [0, 0])
name_at = {e[0]: e[2] for e in win}
for g, a, b in gaps:
    k = name_at.get(b, "?")
    k = k[k.find("shadow::") + 8:] if "shadow::" in k else k
    k = k[:70]
    after[k][0] += g; after[k][1] += 1
end_at = {}
for e in win: end_at[e[1]] = (e[2], e[3])
start_at = {e[0]: (e[2], e[3]) for e in win}
short = lambda k: (k[k.find("shadow::") + 8:] if "shadow::" in k else k)[:60]
pair = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    pk, pq = end_at.get(a, ("?", "?")); nk, nq = start_at.get(b, ("?", "?"))
    pair[(short(pk), pq, short(nk), nq)][0] += g; pair[(short(pk), pq, short(nk), nq)][1] += 1
own = collections.defaultdict(lambda: [0, 0])
is_lib = lambda k: any(x in k for x in ("at::native", "at::cuda", "rocprim", "rocclr", "Cijk_"))
for e in win:
    key = (("torch/rocclr/rocprim " if is_lib(e[2]) else "") + short(e[2]), e[3])
    own[key][0] += e[1] - e[0]; own[key][1] += 1
lib = sum(v[0] for k, v in own.items() if k[0].startswith("torch/"))
print(f"library (torch / rocclr / rocprim) kernels: {lib / nsteps / 1e3:.1f} us per step in {sum(v[1] for k, v in own.items() if k[0].startswith('torch/')) / nsteps:.1f} launches per step")
print("kernels in the window by time (us per step, launches per step, queue):")
for (k, q), (t, c) in sorted(own.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"  {t / nsteps / 1e3:8.1f} us  {c / nsteps:5.1f}/step  [{q}] {k}")
print("largest idle gaps by (kernel that ended [queue] -> kernel that started [queue]):")
for (pk, pq, nk, nq), (g, c) in sorted(pair.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {g / nsteps / 1e3:8.1f} us  {c / nsteps:5.1f}/step  {pk} [{pq}] -> {nk} [{nq}]")
print("idle time by the kernel that follows the gap (us per step, gaps per step):")
for k, (g, c) in sorted(after.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {g / nsteps / 1e3:8.1f} us  {c / nsteps:5.1f}  {k}")
PY
find "$OUT/trace" -name "*kernel_trace.csv" -size +30M -delete
