export PYTHONPATH=$PWD
run() { echo "== $*"; env "$@" timeout 200 python scripts/probe_sampler_batch.py 1024 8192 2>&1 | grep "B=\|rounds"; }
run A=1
run SHADOW_SG_CAPM=1536
run SHADOW_SG_CAPM=1280
run SHADOW_SG_CAPM=1024
run A=1
