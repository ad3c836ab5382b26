# full GPU suite on the product library and on the EXPERIMENTS=1 build, smoke
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ag; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
CRT_LIB=$GRAFT_REPO_ROOT/variants/exp/libcrt.so CRT_EXPERIMENTS_BUILD=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_native_library_is_the_one_running > $O/pytest_exp.log 2>&1; echo "pytest exp rc=$?"; tail -3 $O/pytest_exp.log
