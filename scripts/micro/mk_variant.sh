#!/bin/bash
# A same-box A/B variant of the working tree: _ab/<name>/ = the package, bench.py and the headers, its library built with extra
# hipcc flags (SHADOW_HIPCC_FLAGS).  _ab/ is git-ignored but travels with gpurun: `python _ab/<name>/bench.py ...` on the GPU box.
#   usage: scripts/micro/mk_variant.sh <name> [hipcc flags...]
set -e
cd "$(dirname "$0")/../.."
name="$1"; shift
dst="_ab/$name"
rm -rf "$dst"; mkdir -p "$dst/oracle"
cp -r shadow_gnn_amd include bench.py __graft_entry__.py "$dst/"
cp oracle/__init__.py oracle/sampler_oracle.py oracle/sampler_oracle.c "$dst/oracle/" 2>/dev/null || true
rm -rf "$dst/shadow_gnn_amd/csrc/_obj" "$dst/shadow_gnn_amd/libshadow_hip.so" "$dst/shadow_gnn_amd/__pycache__"
(cd "$dst" && SHADOW_HIPCC_FLAGS="$*" python __graft_entry__.py 2>&1 | grep -E "error|\[build\] ok" )
