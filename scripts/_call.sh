set -e
timeout -k 10 120 python scripts/ba_batch_profile.py 2 20 2000 kernels 2>/dev/null
