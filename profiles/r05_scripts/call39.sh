set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5an; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "larger_than_the_infinity_cache or config4_with_mirror or scheduling_and_loop" --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
