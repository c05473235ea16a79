export PYTHONPATH=$PWD
for pad in 1 9 17 25 33 49; do echo "== pad+1=$pad"; SHADOW_SG_SEG_PAD=$pad timeout 200 python scripts/probe_sampler_batch.py 512 1024 8192 2>&1 | grep "B="; done
echo "== window pad 25"; SHADOW_SG_SCAN_IMPL=window SHADOW_SG_SEG_PAD=25 timeout 200 python scripts/probe_sampler_batch.py 1024 2>&1 | grep "B="
SHADOW_SG_SEG_PAD=25 timeout 600 python -m pytest tests/test_sampler_gpu.py -x -q -m gpu 2>&1 | tail -2
