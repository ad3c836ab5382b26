set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ap; mkdir -p $O
python3 tools/lane_hist.py mesh1m 1 distinct 2>/dev/null | tee $O/distinct_d1.txt
python3 tools/lane_hist.py mesh1m 4 distinct 2>/dev/null | tee $O/distinct_d4.txt
python3 tools/lane_hist.py cornell 1 distinct 2>/dev/null | tee $O/distinct_cornell.txt
