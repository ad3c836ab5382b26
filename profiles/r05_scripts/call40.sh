set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ao; mkdir -p $O
timeout -k 10 1100 python tools/soak.py 5000 2025 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -4 $O/soak.txt
