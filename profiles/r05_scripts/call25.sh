# a wave-wide shared triangle step in the first segment's plain shadow loop (variants share16 / share8), 8 samples per launch
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5z; mkdir -p $O
CRT_LIB=$GRAFT_REPO_ROOT/variants/share16/libcrt.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_native_library_is_the_one_running > $O/pytest_share16.log 2>&1; echo "pytest share16 rc=$?"; tail -3 $O/pytest_share16.log
export AB_LIBS="default,share16,share8"
export AB_CHECK=" "
M="--workload mesh1m"
export AB_RUNS="d1|$M --depth 1 --spp 4;d1_4k|$M --depth 1 --spp 4 --resolution 3840x2160;d1_spp8|$M --depth 1 --spp 8;cornell|--workload cornell --depth 1 --spp 1;d4_inplace|$M --depth 4 --spp 4 --option inplace_shadow=1;hbm|--workload mesh520 --depth 1 --spp 4 --device-built sah"
bash tools/ab_run.sh $O
CRT_LIB=$GRAFT_REPO_ROOT/variants/share16/libcrt.so python3 tools/lane_util.py mesh1m 1 lanes 2>/dev/null | tee $O/lane_share16.txt
