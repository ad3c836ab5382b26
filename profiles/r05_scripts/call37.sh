# L2 hit rate and fabric traffic of the four-segment frame: default / bounce-ray bins / sorted shadow rays (one stream, --pmc passes with the option set)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5al; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in "default|" "bins1|--option ray_bins=1" "sorted|--option sort_shadow=1"; do
  L=${V%%|*}; A=${V#*|}
  ARGS="--no-cpu-baseline --no-live-pmc --no-oracle-check --settle-ms 0 --steps 5 --warmup 2 --streams 1 --workload mesh1m --depth 4 --spp 4 $A"
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d $O/$L/pmc_fetch_mesh1m_d4 -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || exit 1
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $O/$L/pmc_write_mesh1m_d4 -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || exit 1
  python3 - $R $O/$L $L <<'PY'
import sys,os
sys.path.insert(0, sys.argv[1]+'/tools')
import pmc_traffic as pt
d=sys.argv[2]
e=pt.entry_from_dirs({"fetch":d+"/pmc_fetch_mesh1m_d4","write":d+"/pmc_write_mesh1m_d4"},"mesh1m_d4",tail=40)
print(sys.argv[3], {k:e[k] for k in ("dispatches","FETCH_SIZE_KB_per_launch","WRITE_SIZE_KB_per_launch","l2_fabric_bytes_per_launch","l2_hit_rate")})
PY
done | tee $O/table.txt
