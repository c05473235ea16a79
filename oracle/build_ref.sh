#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- never imported by the product path.
#
# Builds the *reference's own* C++ sampler (facebookresearch/shaDow_GNN,
# para_graph_sampler/graph_engine/backend/{ParallelSampler,Graph}.cpp) from the
# sources where they lie under /root/reference into oracle/_ref/ (git-ignored;
# no reference source is copied into this repository).  The reference's CMake
# path is NOT used (its pybind11/ submodule directory is empty); the two source
# files are compiled directly against the image's pip pybind11 headers, which is
# the same dependency (pybind11) the reference pins in its README.
#
# Output: oracle/_ref/ParallelSampler$(python3-config --extension-suffix)
# Used by: oracle/gen_golden.py (golden vectors), tests (oracle pinning),
#          bench.py cpu_baseline leg (kind = "reference").
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${SHADOW_REFERENCE_ROOT:-/root/reference}"
SRC="$REF/para_graph_sampler/graph_engine/backend"
OUT="$HERE/_ref"
if [ ! -f "$SRC/ParallelSampler.cpp" ]; then
  echo "[build_ref] reference sources not present at $SRC -- keeping prebuilt files" >&2
  exit 0
fi
mkdir -p "$OUT"
SUFFIX="$(python3-config --extension-suffix)"
TARGET="$OUT/ParallelSampler$SUFFIX"
if [ "$TARGET" -nt "$SRC/ParallelSampler.cpp" ] && [ "$TARGET" -nt "$SRC/Graph.cpp" ]; then
  echo "[build_ref] up to date: $TARGET"; exit 0
fi
# -std=c++14: the reference uses std::random_shuffle (removed in C++17).
# -DNDEBUG mirrors the reference's Release build (asserts compiled out).
g++ -O3 -DNDEBUG -fPIC -shared -fopenmp -std=c++14 \
    $(python3 -m pybind11 --includes) \
    "$SRC/ParallelSampler.cpp" "$SRC/Graph.cpp" -o "$TARGET"
echo "[build_ref] built $TARGET"
