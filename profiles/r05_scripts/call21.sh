set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5v
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5v/pytest_default.log 2>&1; echo "default pytest rc=$?"; tail -3 gpurun_out/r5v/pytest_default.log
CRT_LIB=$GRAFT_REPO_ROOT/variants/exp/libcrt.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_bench_line_contract --deselect tests/test_gpu_parity.py::test_bench_self_launch_under_rccl_on_one_gpu --deselect tests/test_gpu_parity.py::test_bench_one_process_several_devices > gpurun_out/r5v/pytest_experiments.log 2>&1; echo "experiments pytest rc=$?"; tail -3 gpurun_out/r5v/pytest_experiments.log
timeout -k 10 600 python tools/soak.py 400 501 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5v/soak.txt | tail -4
