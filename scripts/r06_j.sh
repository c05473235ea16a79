#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"
O=$R/gpurun_out/r06j; mkdir -p $O
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 2000 python -m pytest tests/test_sampler_gpu.py tests/test_minibatch_gpu.py tests/test_ref_configs_gpu.py -q -m gpu -x > $O/tests_sampler.log 2>&1; tail -4 $O/tests_sampler.log; grep -n "^E " $O/tests_sampler.log | head -10
python scripts/fuzz_sampler_extreme.py > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
for w in products-khop3-gat5 arxiv-khop-gcn3; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-tail > $O/$w.json 2> $O/$w.err; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06j/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    print(f.split("/")[-1], d["ms_per_step"], "host", d["host_busy_ms_per_step"], "sampler_alone", d["sampler_alone"], {k: v["avg_ms"] for k, v in d["kernels"].items() if k.startswith("sg_")})
PY
