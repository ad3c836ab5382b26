# what regrouping does to the memory side: L2 hit rate and fabric traffic of the four-segment frame with bounce-ray bins / sorted shadow rays
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ak; mkdir -p $O
for V in "default|" "bins1|--option ray_bins=1" "bins4|--option ray_bins=4" "sorted|--option sort_shadow=1"; do
  L=${V%%|*}; A=${V#*|}
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extra --workload mesh1m --depth 4 --spp 4 --steps 20 --warmup 5 $A > $O/$L.json 2> $O/$L.log
  python3 - $L $O/$L.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
print(f"{sys.argv[1]:8s} {d['value']:8.1f} Mray/s  traffic {r.get('traffic')}  l2_hit {r.get('l2_hit_rate')}  issue_busy {r.get('issue_busy')} lane_util {r.get('lane_util')}  ok {d.get('sum_rows_match_oracle')}")
PY
done | tee $O/table.txt
