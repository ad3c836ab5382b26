#!/bin/bash
# is the group phase of walk_batch bound by the texture-address path (8 lanes load the same 80-byte node)?
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --workload mesh1m --depth 4 --spp 4 --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --no-oracle-check --settle-ms 0 --streams 1"
for V in l8 l1; do
  OPT=""; [ $V = l1 ] && OPT="--option lanes_per_ray=1"
  rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta_$V -- $B $OPT > $OUT/ta_$V.json 2> $OUT/ta_$V.log || echo "ta $V failed"
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/sq_$V -- $B $OPT > $OUT/sq_$V.json 2> $OUT/sq_$V.log || echo "sq $V failed"
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcp_$V -- $B $OPT > $OUT/tcp_$V.json 2> $OUT/tcp_$V.log || echo "tcp $V failed"
done
python3 $R/tools/pmc_counters.py $OUT/ta_l8 $OUT/ta_l1 $OUT/sq_l8 $OUT/sq_l1 $OUT/tcp_l8 $OUT/tcp_l1 | grep -v STATS | grep "k_segment<INPLACE>" > $OUT/summary.txt
cat $OUT/summary.txt
