set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5g
for V in refill wavefront def; do
  OPT="--option bounce_refill=1"; [ $V = wavefront ] && OPT="--option bounce_refill=1 --option inplace_shadow=2 --option shadow_pool=128 --option shadow_refill_min=16"
  [ $V = def ] && OPT="--option inplace_shadow=2 --option shadow_pool=128 --option shadow_refill_min=16"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5g/prof_$V -- python3 $GRAFT_REPO_ROOT/bench.py --workload mesh1m --depth 4 --spp 4 --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-live-pmc --no-oracle-check $OPT > $GRAFT_REPO_ROOT/gpurun_out/r5g/prof_$V.log 2>&1)
  echo "== $V"; find gpurun_out/r5g/prof_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -6 {} | cut -c1-200'
done
python3 tools/lane_util.py --help 2>&1 | head -5
