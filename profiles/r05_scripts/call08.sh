set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
for O in "" "bounce_refill=1" "bounce_refill=1 refill_pool=128" "bounce_refill=1 refill_min=32" "bounce_refill=1 refill_min=65 refill_pool=64" "inplace_shadow=2" "inplace_shadow=2 shadow_pool=128 shadow_refill_min=16" "inplace_shadow=2 shadow_pool=256 shadow_refill_min=8"; do
  python3 tools/lane_util.py mesh1m 4 $O 2>/dev/null | tee -a gpurun_out/r5h/lane_util.txt
done
