#!/bin/bash
# probe: what does regrouping the bounce rays by direction octant alone (or octant + 2^3 cells) give the walks?  existing ray_bins machinery, coarser keys
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4x; mkdir -p $OUT
cd $R
for V in "full|" "oct|-DCRT_BIN_KEY_MODE=1" "oct8|-DCRT_BIN_KEY_MODE=2"; do
  L=${V%%|*}; EX=${V#*|}
  rm -f caitlynrenderer_amd/csrc/rt_kernels.o; make -C caitlynrenderer_amd/csrc -s EXTRA="$EX" > /dev/null 2>&1
  for B in 0 1 5; do
    echo "== key $L ray_bins $B" >> $OUT/lane_util.txt
    timeout -k 10 300 python tools/lane_util.py mesh1m 4 ray_bins=$B >> $OUT/lane_util.txt 2>&1
  done
done
export AB_BUILDS="full|;oct|-DCRT_BIN_KEY_MODE=1;oct8|-DCRT_BIN_KEY_MODE=2"
export AB_RUNS="d4|--workload mesh1m --depth 4 --spp 4;d4_b1|--workload mesh1m --depth 4 --spp 4 --option ray_bins=1;d4_b5|--workload mesh1m --depth 4 --spp 4 --option ray_bins=5;hbm_d4_b1|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10 --option ray_bins=1"
bash tools/ab.sh $OUT
grep -v amdgpu.ids $OUT/lane_util.txt
