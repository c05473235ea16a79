#!/bin/bash
# A/B on one box: the GEMM-epilogue kernels' A operand through non-temporal loads (FUSED_A_NT) -- does the Zs tile a workgroup
# parks at the phase boundary then survive in L2 until the epilogue reads it back?  Headline step + per-launch times.
cd "$(dirname "$0")/../.."
export PYTHONPATH=.
cp shadow_gnn_amd/libshadow_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for v in "" "-DFUSED_A_NT"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $v -Iinclude -Ishadow_gnn_amd/csrc \
      shadow_gnn_amd/csrc/gemm_fused.hip -o /tmp/gemm_fused_ab.o || exit 1
  objs=$(ls shadow_gnn_amd/csrc/_obj/*.o | grep -v gemm_fused.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_fused_ab.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-tail > /tmp/ab.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
k = d['kernels']
print('variant [$v] rep $rep: ms/step', d['ms_per_step'], {x: k[x]['avg_ms'] for x in k if x.startswith('gemm_act_norm_fwd') or x.startswith('gemm_an_bwd')})
PY
done
done
cp /tmp/lib_orig.so shadow_gnn_amd/libshadow_hip.so
