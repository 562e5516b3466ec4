set -e
timeout -k 10 600 python -m pytest tests/test_ba_gpu.py -m gpu -x -q 2>&1 | tail -2
ORBX_BA_TIMING=1 timeout -k 10 120 python scripts/ba_batch_profile.py 32 20 2000 2>&1 | tail -5
echo "== no split"
ORBX_BA_NO_SPLIT=1 ORBX_BA_TIMING=1 timeout -k 10 120 python scripts/ba_batch_profile.py 32 20 2000 2>&1 | tail -3
for i in 1 2 3; do timeout -k 10 120 python scripts/ba_batch_profile.py 32 20 2000 2>&1 | tail -1; done
