set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5am; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default"
export AB_CHECK=" "
export AB_RUNS="d1|$M --depth 1;d1_all|$M --depth 1 --option inplace_shadow=0;d1_all_ws0|$M --depth 1 --option inplace_shadow=0 --option wave_samples=0"
bash tools/ab_run.sh $O
python3 - <<'PY'
import json
for n in ("d1","d1_all","d1_all_ws0"):
    d=json.loads([l for l in open(f"gpurun_out/r5am/default_{n}.json") if l.startswith('{')][-1]); print(n, d['config'].get('launch'))
PY
