#!/bin/bash
# A/B of kernel builds on the GPU box.  usage: tools/ab.sh OUTDIR  (reads build variants from $AB_BUILDS, a ';'-separated list of
# "label|EXTRA defines", and runs from $AB_RUNS, a ';'-separated list of "label|bench.py arguments"); appends one line per (build, run):
#   build run value(Mray/s) ms_per_step launch_ms
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$1; mkdir -p $OUT
IFS=';' read -ra BUILDS <<< "$AB_BUILDS"
IFS=';' read -ra RUNS <<< "$AB_RUNS"
for B in "${BUILDS[@]}"; do
  BL=${B%%|*}; EX=${B#*|}
  rm -f $R/caitlynrenderer_amd/csrc/rt_kernels.o $R/caitlynrenderer_amd/csrc/crt_device.o
  make -C $R/caitlynrenderer_amd/csrc -s EXTRA="$EX" > $OUT/build_$BL.log 2>&1 || { echo "build $BL failed"; tail -5 $OUT/build_$BL.log; continue; }
  for RUN in "${RUNS[@]}"; do
    RL=${RUN%%|*}; ARGS=${RUN#*|}
    python3 $R/bench.py --gpus 1 --no-cpu-baseline --no-live-pmc --no-oracle-check --steps ${AB_STEPS:-30} --warmup 5 $ARGS > $OUT/${BL}_$RL.json 2> $OUT/${BL}_$RL.log
    python3 - "$BL" "$RL" $OUT/${BL}_$RL.json <<'PY' | tee -a $OUT/table.txt
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    print(f"{sys.argv[1]:14s} {sys.argv[2]:18s} {d['value']:10.1f} Mray/s  {d['ms_per_step']:8.4f} ms/step  launch {d['roofline']['launch_ms']:.4f} ms  streams {d['config'].get('streams')}  form {d['config'].get('launch')}")
except Exception as e:
    print(f"{sys.argv[1]:14s} {sys.argv[2]:18s} FAILED {e!r}")
PY
  done
done
# leave the default build behind
rm -f $R/caitlynrenderer_amd/csrc/rt_kernels.o $R/caitlynrenderer_amd/csrc/crt_device.o
make -C $R/caitlynrenderer_amd/csrc -s > /dev/null 2>&1
