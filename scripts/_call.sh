set -e -o pipefail
bash scripts/ab_bench.sh r03y_ab build_ab/base.so build_ab/hs2.so
