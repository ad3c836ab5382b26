#!/bin/bash
# uniform steps in every first-segment kernel + traverse_pool without a carried flag: parity, crt_trace throughput, A/B on the small-scene cases
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4aj; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; RC=$?; echo "pytest rc $RC"; tail -5 $OUT/pytest.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python tools/refill_probe.py mesh1m > $OUT/refill_probe.txt 2>&1; grep -v amdgpu $OUT/refill_probe.txt | grep "pool  64\|pool 128 refill_min 16\|pool 256 refill_min 16" | head -12
export AB_BUILDS="dflt|;no_single|-DCRT_UNIFORM_SINGLE=0 -DCRT_UNIFORM_PLAIN=0"
export AB_RUNS="cornell|--workload cornell --depth 1 --spp 1 --steps 200;cornell4|--workload cornell --depth 1 --spp 4 --steps 100;cornell_d4|--workload cornell --depth 4 --spp 1 --steps 100;d1_spp1|--workload mesh1m --depth 1 --spp 1 --steps 60;d4_spp1|--workload mesh1m --depth 4 --spp 1 --steps 30;d1|--workload mesh1m --depth 1 --spp 4"
bash tools/ab.sh $OUT
