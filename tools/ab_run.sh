#!/bin/bash
# A/B of prebuilt library variants on the GPU box (tools/variant.sh builds them beforehand, on the CPU container).
#   usage: AB_LIBS="label[,label...]" AB_RUNS="label|bench.py arguments;..." tools/ab_run.sh OUTDIR
# `default` = the product library caitlynrenderer_amd/libcrt.so, any other label = variants/<label>/libcrt.so (through CRT_LIB).
# Appends one line per (library, run) to OUTDIR/table.txt: library run Mray/s ms_per_step launch_ms
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; mkdir -p "$OUT"
IFS=',' read -ra LIBS <<< "${AB_LIBS:-default}"
IFS=';' read -ra RUNS <<< "$AB_RUNS"
for RUN in "${RUNS[@]}"; do
  RL=${RUN%%|*}; ARGS=${RUN#*|}
  for BL in "${LIBS[@]}"; do
    unset CRT_LIB_ABI; [ -f "$R/variants/$BL/abi" ] && export CRT_LIB_ABI=$(cat "$R/variants/$BL/abi")
    if [ "$BL" = default ]; then unset CRT_LIB; else export CRT_LIB="$R/variants/$BL/libcrt.so"; [ -f "$CRT_LIB" ] || { echo "$BL: no such variant"; continue; }; fi
    timeout -k 10 ${AB_TIMEOUT:-240} python3 "$R/bench.py" --gpus 1 --no-cpu-baseline --no-live-pmc ${AB_CHECK:---no-oracle-check} --steps ${AB_STEPS:-30} --warmup 5 $ARGS > "$OUT/${BL}_$RL.json" 2> "$OUT/${BL}_$RL.log"
    python3 - "$BL" "$RL" "$OUT/${BL}_$RL.json" <<'PY' | tee -a "$OUT/table.txt"
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    print(f"{sys.argv[1]:18s} {sys.argv[2]:16s} {d['value']:10.1f} Mray/s  {d['ms_per_step']:8.4f} ms/step  launch {d['roofline']['launch_ms']:.4f} ms  streams {d['config'].get('streams')}  ok {d.get('sum_rows_match_oracle')}")
except Exception as e:
    print(f"{sys.argv[1]:18s} {sys.argv[2]:16s} FAILED {e!r}")
PY
  done
done
unset CRT_LIB CRT_LIB_ABI
