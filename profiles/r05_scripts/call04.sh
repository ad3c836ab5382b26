set -o pipefail
cd $GRAFT_REPO_ROOT
export AB_LIBS="default,r4,default,r4"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;d4|--workload mesh1m --depth 4 --spp 4;cornell|--workload cornell --depth 1 --spp 1"
bash tools/ab_run.sh gpurun_out/r5d
export TMPDIR=/tmp
for V in inplace def; do
  OPT=""; [ $V = def ] && OPT="--option inplace_shadow=2 --option shadow_pool=128 --option shadow_refill_min=16"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5d/prof_$V -- python3 $GRAFT_REPO_ROOT/bench.py --workload mesh1m --depth 4 --spp 4 --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-live-pmc --no-oracle-check $OPT > $GRAFT_REPO_ROOT/gpurun_out/r5d/prof_$V.log 2>&1)
  find gpurun_out/r5d/prof_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
done
