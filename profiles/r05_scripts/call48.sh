# the 8 M-triangle tree is 14 levels deep: 13 LDS stack entries per lane = 7.5 KB per wave = 5.25 waves per SIMD.  What would fewer LDS entries buy
# (valid only while stack_overflows stays 0)?
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5aw; mkdir -p $O
H="--workload mesh520 --spp 4 --device-built sah"
export AB_LIBS="default"
export AB_CHECK=" "
R=""
for E in 13 10 8 7 6; do R="$R;hbm_d4_e$E|$H --depth 4 --option stack_entries=$E"; done
for E in 13 8; do R="$R;hbm_d1_e$E|$H --depth 1 --option stack_entries=$E"; done
export AB_RUNS="${R#;}"
bash tools/ab_run.sh $O
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5aw/default_hbm_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], d['config'].get('stack_overflows'), d.get('sum_rows_match_oracle'))
PY
