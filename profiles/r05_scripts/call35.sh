# address-translation counters of the four-segment frame (is the L1 TLB a limiter for incoherent rays?)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5aj; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-live-pmc --no-oracle-check --settle-ms 0 --steps 5 --warmup 2 --streams 1 --workload mesh1m --depth 4 --spp 4"
timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d $O/pass1 -- python3 $R/bench.py $ARGS > $O/pass1.json 2> $O/pass1.log; echo rc=$?
timeout -k 10 150 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum --output-format csv -d $O/pass2 -- python3 $R/bench.py $ARGS > $O/pass2.json 2> $O/pass2.log; echo rc=$?
ARGS="--no-cpu-baseline --no-live-pmc --no-oracle-check --settle-ms 0 --steps 5 --warmup 2 --streams 1 --workload mesh520 --depth 4 --spp 4 --device-built sah"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d $O/pass3 -- python3 $R/bench.py $ARGS > $O/pass3.json 2> $O/pass3.log; echo rc=$?
python3 $R/tools/pmc_anatomy.py $O 40 | grep -v "^   -> " | tee $O/anatomy.txt | grep "^==\|UTCL1\|cycles"
