#!/bin/bash
# re-entry check of HEAD: full GPU suite, then the pending A/B (first-segment shadow walk = plain loop + group phase)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4v; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -3 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
export AB_BUILDS="dflt|;first_any_groups|-DCRT_FIRST_ANY_GROUPS=1;wide_loads|-DCRT_GROUP_WIDE_LOADS=1"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160;hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10;d4|--workload mesh1m --depth 4 --spp 4;cornell|--workload cornell --depth 1 --spp 1 --steps 200"
bash tools/ab.sh $OUT
