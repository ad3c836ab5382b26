set -o pipefail
cd $GRAFT_REPO_ROOT
export AB_LIBS="default,r4,default,r4"
export AB_RUNS="d1|--workload mesh1m --depth 1 --spp 4;cornell|--workload cornell --depth 1 --spp 1;d4|--workload mesh1m --depth 4 --spp 4;d4_def|--workload mesh1m --depth 4 --spp 4 --option inplace_shadow=2 --option shadow_pool=128 --option shadow_refill_min=16"
bash tools/ab_run.sh gpurun_out/r5f
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5f/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5f/pytest.log
