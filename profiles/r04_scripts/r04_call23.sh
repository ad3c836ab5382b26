#!/bin/bash
# probe: share of node steps whose enabled lanes all fetch the same node (and share the octant)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4y; mkdir -p $OUT
cd $R
for V in "all|" "same_node|-DCRT_HIST_UNIFORM=1" "same_node_oct|-DCRT_HIST_UNIFORM=2"; do
  L=${V%%|*}; EX=${V#*|}
  rm -f caitlynrenderer_amd/csrc/rt_kernels.o; make -C caitlynrenderer_amd/csrc -s EXTRA="$EX" > /dev/null 2>&1
  for D in 1 4; do
    echo "== $L depth $D" >> $OUT/uniform.txt
    timeout -k 10 300 python tools/lane_hist.py mesh1m $D 2>&1 | grep -v "amdgpu.ids\|lanes per ray when" >> $OUT/uniform.txt
  done
done
rm -f caitlynrenderer_amd/csrc/rt_kernels.o; make -C caitlynrenderer_amd/csrc -s > /dev/null 2>&1
cat $OUT/uniform.txt
