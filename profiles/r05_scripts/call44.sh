set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5as; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_issue_cycles tools/ubench/valu_issue_cycles.hip > $O/build.log 2>&1 && timeout -k 10 300 /tmp/valu_issue_cycles > $O/valu_issue_cycles.txt 2>&1; echo rc=$?
grep "4 waves/SIMD" $O/valu_issue_cycles.txt | head -30
