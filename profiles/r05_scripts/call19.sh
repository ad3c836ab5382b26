set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5t
for V in def refill; do
  OPT=""; [ $V = refill ] && OPT="--option bounce_refill=1 --option refill_min=8"
  for P in sq fetch; do
    C="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; [ $P = fetch ] && C="FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
    echo "pass $V $P"
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5t/${P}_$V -- python3 $GRAFT_REPO_ROOT/bench.py --workload mesh1m --depth 4 --spp 4 --steps 4 --warmup 1 --streams 1 --settle-ms 0 --no-cpu-baseline --no-live-pmc --no-oracle-check $OPT > $GRAFT_REPO_ROOT/gpurun_out/r5t/${P}_$V.log 2>&1) || echo "pass failed"
  done
done
python3 - <<'PY'
import csv,glob,collections
for V in ('def','refill'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ('sq_','fetch_'):
        for f in glob.glob(f'gpurun_out/r5t/{d}{V}/**/*counter_collection.csv',recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==',V)
    for k,c in agg.items():
        if 'k_segment' not in k and 'k_closest' not in k and 'k_shadow' not in k: continue
        n=len(c['SQ_INSTS_VALU'])
        if n<6: continue
        def top(name):
            v=sorted(c[name]); return sum(v[len(v)//2:])/max(1,len(v)-len(v)//2) if v else 0
        insts=top('SQ_INSTS_VALU'); act=top('SQ_ACTIVE_INST_VALU'); thr=top('SQ_THREAD_CYCLES_VALU'); gui=top('GRBM_GUI_ACTIVE')/8; waves=top('SQ_WAVES')
        busy=2*insts/(1024*gui) if gui else 0
        print(f"{k:72s} n={n:3d} VALU/launch {insts/1e6:7.1f}M waves {waves:8.0f} instr/wave {insts/max(1,waves):7.0f} lane_util {thr/max(1,64*act):.3f} issue_busy {busy:.3f} cycles {gui/1e3:7.0f}k fetchMB {top('FETCH_SIZE')*2/1024:8.1f} wait_any/wave_cycles {top('SQ_WAIT_ANY')/max(1,top('SQ_WAVE_CYCLES')):.3f}")
PY
