set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scheduling_and_loop or deferred_shadow or render_frames_equals" > gpurun_out/r5r/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5r/pytest.log
export AB_LIBS="default"
M="--workload mesh1m --depth 4 --spp 4"
export AB_RUNS="d4_sorted|$M;d4_unsorted|$M --option sort_shadow=0;d4_sorted_p128|$M --option shadow_pool=128;d4_sorted_r8|$M --option shadow_refill_min=8;d4_sorted_ls|$M --option shadow_refill_min=65 --option shadow_pool=64;d2_sorted|--workload mesh1m --depth 2 --spp 4;d2_unsorted|--workload mesh1m --depth 2 --spp 4 --option sort_shadow=0;hbm4_sorted|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10;hbm4_unsorted|--workload mesh520 --device-built sah --depth 4 --spp 4 --steps 10 --option sort_shadow=0"
bash tools/ab_run.sh gpurun_out/r5r
python3 tools/lane_util.py mesh1m 4 2>/dev/null | tail -5
