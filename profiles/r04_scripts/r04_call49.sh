#!/bin/bash
# waves per CU of the one-pass headline kernel (71 VGPRs: 28 fit): taken off with unused LDS (1,280-byte units; 5,120 bytes per wave at 1 M triangles)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4ax; mkdir -p $OUT
cd $R
for PAD in 0 1280 2560; do
  for RUN in "d1|--workload mesh1m --depth 1 --spp 4" "k4|--workload mesh1m --depth 1 --spp 4 --resolution 3840x2160" "hbm_d1|--workload mesh520 --device-built sah --depth 1 --spp 4 --steps 10" "d1_spp1|--workload mesh1m --depth 1 --spp 1 --steps 60"; do
    RL=${RUN%%|*}; ARGS=${RUN#*|}
    CRT_LDS_PAD=$PAD python3 bench.py --gpus 1 --no-cpu-baseline --no-live-pmc --no-oracle-check --steps 30 --warmup 5 $ARGS > $OUT/pad${PAD}_$RL.json 2> $OUT/pad${PAD}_$RL.log
    python3 -c "
import json,sys
d=json.loads([l for l in open('$OUT/pad${PAD}_$RL.json') if l.startswith('{')][-1]); print('pad $PAD', '$RL', d['value'], d['ms_per_step'])"
  done
done
