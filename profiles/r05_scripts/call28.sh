# does the oracle check after the clock change the timed figure?  (it must not)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ac; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default,shocc8"
export AB_RUNS="d4_nocheck|$M --depth 4;d2_nocheck|$M --depth 2"
bash tools/ab_run.sh $O
export AB_CHECK=" "
export AB_RUNS="d4_check|$M --depth 4;d2_check|$M --depth 2"
bash tools/ab_run.sh $O
unset AB_CHECK
export AB_RUNS="d4_nocheck2|$M --depth 4"
bash tools/ab_run.sh $O
rocm-smi --showclocks 2>/dev/null | head -20
