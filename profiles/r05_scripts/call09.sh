set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scheduling_and_loop or deferred_shadow or render_frames_equals" > gpurun_out/r5i/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5i/pytest.log
export AB_LIBS="default"
M="--workload mesh1m --depth 4 --spp 4"
export AB_RUNS="d4|$M;d4_def|$M --option inplace_shadow=2 --option shadow_pool=128 --option shadow_refill_min=16;d4_defP16|$M --option inplace_shadow=2 --option persistent=1 --option shadow_refill_min=16;d4_defP32|$M --option inplace_shadow=2 --option persistent=1 --option shadow_refill_min=32;d4_defP8|$M --option inplace_shadow=2 --option persistent=1 --option shadow_refill_min=8;d4_wfP16|$M --option inplace_shadow=2 --option persistent=1 --option shadow_refill_min=16 --option bounce_refill=1 --option refill_min=16;d4_wfP32|$M --option inplace_shadow=2 --option persistent=1 --option shadow_refill_min=16 --option bounce_refill=1 --option refill_min=32;d4_wfP8|$M --option inplace_shadow=2 --option persistent=1 --option shadow_refill_min=16 --option bounce_refill=1 --option refill_min=8;d4_refP16|$M --option persistent=1 --option bounce_refill=1 --option refill_min=16"
bash tools/ab_run.sh gpurun_out/r5i
