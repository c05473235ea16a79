#!/bin/bash
# A/B on one box: gat_row_bwd_kernel (106 VGPRs, 4 wavefronts per SIMD) under a register cap for 5 / 6 / 8 resident wavefronts.
cd "$(dirname "$0")/../.."
export PYTHONPATH=.
cp shadow_gnn_amd/libshadow_hip.so /tmp/lib_orig.so
for v in ${VARIANTS:-"" "-DSHADOW_GAT_ROW_BWD_WAVES=5" "-DSHADOW_GAT_ROW_BWD_WAVES=6" "-DSHADOW_GAT_ROW_BWD_WAVES=8" "-DSHADOW_GAT_GROUPS_ROW=2_-DSHADOW_GAT_ROW_BWD_WAVES=8" ""}; do v=${v//_-D/ -D}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $v -Iinclude -Ishadow_gnn_amd/csrc \
      shadow_gnn_amd/csrc/gat.hip -o /tmp/gat_ab.o -Rpass-analysis=kernel-resource-usage 2> /tmp/gat_ab.log || exit 1
  regs=$(grep -A6 "gat_row_bwd_kernelILi64E" /tmp/gat_ab.log | grep -E " VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' ')
  objs=$(ls shadow_gnn_amd/csrc/_obj/*.o | grep -v gat.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gat_ab.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  python bench.py --workload products-khop3-gat5 --steps 30 --warmup 6 --no-cpu-baseline --no-tail > /tmp/ab.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
k = d['kernels']
print('variant [$v] $regs: ms/step', d['ms_per_step'], ' gat_bwd', k['gat_bwd_F256_H4']['avg_ms'], ' gat_fwd', k['gat_fwd_F256_H4']['avg_ms'])
PY
done
cp /tmp/lib_orig.so shadow_gnn_amd/libshadow_hip.so
