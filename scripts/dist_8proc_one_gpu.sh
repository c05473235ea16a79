#!/usr/bin/env bash
#  bench.py as the driver launches it for 8 GPUs -- eight ranks through torch.distributed.run -- on a box
# with ONE GPU: SHADOW_DIST_BACKEND=gloo lets all ranks share cuda:0.  NOT a scaling number (eight processes time-slice one
# GPU and the gradient exchange goes through host memory); what it records is the host side of an 8-rank node: eight ranks
# issuing the same collective sequence through GradSync, and per-rank host_busy / own_ms_per_step with the ranks' host
# threads pinned (dist.pin_host_threads) and unpinned, next to the single-process figures of the same batch size.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/dist8"; rm -rf "$OUT"; mkdir -p "$OUT"
export PYTHONPATH="$R" HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$R"
COMMON="--batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-tail --no-other-workloads"
python bench.py --gpus 1 $COMMON > "$OUT/one.json" 2> "$OUT/one.err"; echo "one rc=$?"
for MODE in pinned unpinned; do
  EXTRA=""; [ "$MODE" = unpinned ] && EXTRA="--no-pin"
  SHADOW_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus 8 $COMMON $EXTRA > "$OUT/eight_$MODE.json" 2> "$OUT/eight_$MODE.err"; echo "eight $MODE rc=$?"
done
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
def load(n):
    ls = [l for l in open(os.path.join(out, n)).read().splitlines() if l.startswith("{")]
    return json.loads(ls[-1]) if ls else None
res = {"note": "8 ranks sharing ONE GPU over gloo (functional / host-side evidence, NOT a scaling measurement): products-khop-sage5 at 128 roots "
               "per rank; per_rank figures are ms per step", "host_cpus": os.cpu_count()}
one = load("one.json")
if one:
    res["one_process"] = {k: one[k] for k in ("ms_per_step", "host_busy_ms_per_step", "host_enqueue_ms_per_step", "value")}
for m in ("pinned", "unpinned"):
    d = load(f"eight_{m}.json")
    if d:
        res[f"eight_ranks_{m}"] = dict(ms_per_step=d["ms_per_step"], value=d["value"], n_gpus=d["n_gpus"], global_batch=d["config"]["global_batch"],
                                       nodes_per_step=d["config"]["nodes_per_step"], dist=d["dist"])
json.dump(res, open(os.path.join(out, "r06_dist_8proc_one_gpu.json"), "w"), indent=1)
print(json.dumps(res)[:3000])
PY
