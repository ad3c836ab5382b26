set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
timeout -k 10 1000 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5m/bench.json 2> gpurun_out/r5m/bench.log; echo "bench rc=$?"; tail -3 gpurun_out/r5m/bench.log | cut -c1-300; wc -c gpurun_out/r5m/bench.json
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r5m/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5m/pytest.log
