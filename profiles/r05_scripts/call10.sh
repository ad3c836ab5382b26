set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scheduling_and_loop or deferred_shadow or render_frames_equals" > gpurun_out/r5j/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5j/pytest.log
export AB_LIBS="default"
M="--workload mesh1m --depth 4 --spp 4"
P="--option persistent=1"
export AB_RUNS="d4_def|$M --option inplace_shadow=2 --option shadow_pool=128 --option shadow_refill_min=16;d4_defP256_16|$M --option inplace_shadow=2 $P --option shadow_refill_min=16 --option shadow_pool=256;d4_defP512_8|$M --option inplace_shadow=2 $P --option shadow_refill_min=8 --option shadow_pool=512;d4_defP128_16|$M --option inplace_shadow=2 $P --option shadow_refill_min=16 --option shadow_pool=128;d4_wfP256_8|$M --option inplace_shadow=2 $P --option shadow_refill_min=16 --option shadow_pool=256 --option bounce_refill=1 --option refill_min=8;d4_wfP512_8|$M --option inplace_shadow=2 $P --option shadow_refill_min=16 --option shadow_pool=256 --option bounce_refill=1 --option refill_min=8 --option refill_pool=512;d4_wfP128_16|$M --option inplace_shadow=2 $P --option shadow_refill_min=16 --option shadow_pool=256 --option bounce_refill=1 --option refill_min=16 --option refill_pool=128"
bash tools/ab_run.sh gpurun_out/r5j
