set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5av; mkdir -p $O
( time python bench.py > $O/bench_default_box2.json 2> $O/bench_default_box2.log ) 2> $O/time.txt; echo rc=$?; tail -3 $O/time.txt
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5av/bench_default_box2.json').read())
print(d['value'], d['ms_per_step'], d['step_ms_spread'], d['sum_rows_match_oracle'])
print({k:(v.get('value'),v.get('sum_rows_match_oracle', v.get('rgba_rows_match_oracle'))) for k,v in d['extras'].items()})
PY
