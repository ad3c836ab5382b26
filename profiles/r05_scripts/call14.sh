set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5o/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r5o/pytest.log
