set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5aq; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "histograms or float_planes or special_materials" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
