export PYTHONPATH=$PWD
for cfg in "1024 8192 2048" "1024 4096 2048" "512 4096 1024" "512 8192 2048" "256 4096 1024"; do
  set -- $cfg
  echo "== T=$1 bitwords=$2 capm=$3"
  SHADOW_SG_SCAN_THREADS=$1 SHADOW_SG_BITWORDS=$2 SHADOW_SG_CAPM=$3 python scripts/probe_sampler_batch.py ${SWEEP_B:-1024} 2>&1 | grep -E "scan:|B="
done
