#!/bin/bash
# A/B on one box: the pipelined CSR SpMM's long-row loop with 2 / 4 / 6 / 8 gathers in flight (SHADOW_SPMM_LONG) on a PPR and a k-hop
# batch, beside the block-diagonal LDS kernel on the same batches.
cd "$(dirname "$0")/../.."
export PYTHONPATH=.
cp shadow_gnn_amd/libshadow_hip.so /tmp/lib_orig.so
for m in ppr khop; do echo "block-diagonal kernel, $m: $(METHOD=$m BLOCKDIAG=1 python scripts/probe_spmm.py 256 2>&1 | tail -1)"; done
for v in 2 4 6 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DSHADOW_SPMM_LONG=$v -Iinclude -Ishadow_gnn_amd/csrc shadow_gnn_amd/csrc/aggregate.hip -o /tmp/agg_ab.o || exit 1
  objs=$(ls shadow_gnn_amd/csrc/_obj/*.o | grep -v aggregate.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/agg_ab.o -o shadow_gnn_amd/libshadow_hip.so || exit 1
  for m in ppr khop; do echo "pipelined CSR kernel, $v in flight, $m: $(METHOD=$m BLOCKDIAG=0 python scripts/probe_spmm.py 256 2>&1 | tail -1)"; done
done
cp /tmp/lib_orig.so shadow_gnn_amd/libshadow_hip.so
