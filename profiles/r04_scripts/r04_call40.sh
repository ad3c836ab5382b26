#!/bin/bash
# LDS stack of depth - 1 entries: parity, then A/B on the deep tree (8 M triangles: 21 -> 24 waves per CU) and the others
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ao; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|"
export AB_RUNS="hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;hbm_d4|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;d1|--workload mesh1m --depth 1 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
timeout -k 10 600 python tools/soak.py 300 > $OUT/soak.txt 2>&1; tail -3 $OUT/soak.txt
