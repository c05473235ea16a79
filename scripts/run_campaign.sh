#!/usr/bin/env bash
# Local driver of the round's measurement campaign: clears the LOCAL copies of the campaign's output directories first (gpurun
# merges what the box wrote into gpurun_out/, it does not mirror -- stale per-pid files of an earlier run would be picked up by
# scripts/publish_profiles.py / make_traffic.py), runs scripts/campaign_<tag>.sh on a GPU box, then publishes into profiles/.
#   usage: scripts/run_campaign.sh r04
set -uo pipefail
TAG="${1:-r04}"
cd "$(dirname "$0")/.."
rm -rf "gpurun_out/prof_$TAG" "gpurun_out/prof_${TAG}_w" "gpurun_out/bench_$TAG"
/usr/local/graft/bin/gpurun --timeout 3000 -- "bash scripts/campaign_$TAG.sh" || exit $?
python scripts/publish_profiles.py "gpurun_out/prof_$TAG" "$TAG"
python scripts/make_traffic.py "gpurun_out/prof_$TAG" "$TAG" products-khop-sage5 > /dev/null
for f in gpurun_out/bench_$TAG/*.json; do
  w=$(basename "$f" .json)
  python - "$f" "profiles/${TAG}_bench_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
json.dump(d, open(sys.argv[2], "w"), indent=1)
PY
done
cp gpurun_out/prof_${TAG}_w/${TAG}_kernel_stats_*.csv profiles/
cp "gpurun_out/full_gpu_tests_$TAG.log" "profiles/${TAG}_gpu_tests_tail.txt"
python scripts/design_tables.py "$TAG" --write
