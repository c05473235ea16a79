#!/bin/bash
# A/B of non-temporal vs plain stores on the gather / aggregation outputs (csrc/aggregate.hip), one gpurun call, same box:
# rebuilds aggregate.o per variant and reads the per-kernel times of bench.py's instrumented steps.
cd "$(dirname "$0")/.."
for v in "0 0" "1 1" "0 1" "1 0" "0 0"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DSHADOW_NT_GATHER_OUT=$1 -DSHADOW_NT_SPMM_OUT=$2 -Iinclude -Ishadow_gnn_amd/csrc \
      shadow_gnn_amd/csrc/aggregate.hip -o shadow_gnn_amd/csrc/_obj/aggregate.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC shadow_gnn_amd/csrc/_obj/*.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tail 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('nt_gather=$1 nt_spmm=$2 step_ms', d['ms_per_step'], ' '.join(f\"{n}={k[n]['avg_ms']*1e3:.1f}us\" for n in ('gather_F100','spmm_F100','spmm_F256','gemm_act_norm_fwd_nb2_N256_Ktail','gemm_act_norm_fwd_nb2_N256') if n in k))"
done
