#!/bin/bash
# usage: scratch/pmc.sh <tag> <bench args...>   (runs on the GPU box; one rocprofv3 --pmc pass per counter group)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r GROUP; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/pass$i.json 2> $OUT/pass$i.log || echo "pass $i failed"
  echo "pass $i done: $GROUP"
done <<'GROUPS'
SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY
FETCH_SIZE TCC_HIT_sum
WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
GROUPS
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
