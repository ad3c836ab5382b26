# final tree: full GPU suite, smoke, and the headline / four-segment figures once more
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ar; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
export AB_LIBS="default"; export AB_CHECK=" "
M="--workload mesh1m --spp 4"
export AB_RUNS="d1|$M --depth 1;d4|$M --depth 4"
bash tools/ab_run.sh $O
