#!/bin/bash
# GPU box, round 4, call 1: (a) first-call cost of the build-on-device scene, attributed; (b) where the headline launch's WRITE_SIZE comes from.
# usage: bash tools/r04_probe1.sh   (from the repo root; writes gpurun_out/r4a/)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/build_probe.py > $OUT/build_probe.txt 2>&1
echo "build probe done"; cat $OUT/build_probe.txt | grep -v "^W\|^E" | tail -30
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_after_frame -- python3 $R/tools/build_probe.py after_frame > $OUT/trace_after_frame.log 2>&1
echo "kernel trace done"
B="python3 $R/bench.py --gpus 1 --workload mesh1m --depth 1 --spp 4 --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --settle-ms 0 --streams 1"
for V in default wide0 tiles0; do
  OPT=""
  [ $V = wide0 ] && OPT="--option wide_first=0"
  [ $V = tiles0 ] && OPT="--option adaptive_tiles=0"
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/w1_$V -- $B $OPT > $OUT/w1_$V.json 2> $OUT/w1_$V.log || echo "w1 $V failed"
  rocprofv3 --pmc TCC_WRITE_sum TCC_WRITEBACK_sum TCC_ATOMIC_sum TCC_NORMAL_EVICT_sum --output-format csv -d $OUT/w2_$V -- $B $OPT > $OUT/w2_$V.json 2> $OUT/w2_$V.log || echo "w2 $V failed"
  rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAVES --output-format csv -d $OUT/w3_$V -- $B $OPT > $OUT/w3_$V.json 2> $OUT/w3_$V.log || echo "w3 $V failed"
  echo "variant $V done"
done
python3 $R/tools/pmc_counters.py --skip-last 10 $OUT/w1_* $OUT/w2_* $OUT/w3_* > $OUT/write_counters.txt 2>&1
cat $OUT/write_counters.txt
# keep the pull small: the raw per-dispatch CSVs of the trace are large
find $OUT -name "*kernel_trace.csv" -size +20M -delete
