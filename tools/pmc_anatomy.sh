#!/bin/bash
# Where a kernel's idle issue slots go: small --pmc passes (at most 2-4 counters of one block per pass, each under its own timeout) of one
# bench.py workload, summarised per kernel by tools/pmc_anatomy.py.
#   usage (on the box): bash tools/pmc_anatomy.sh OUTDIR "bench.py arguments" [TAIL]
# TAIL = segment launches of the run's closing single-sample frames to leave out: min(10, steps x spp) x depth (10 for one segment, 40 for four)
R=$GRAFT_REPO_ROOT; O=$1; ARGS="--no-cpu-baseline --no-live-pmc --no-oracle-check --settle-ms 0 --steps 5 --warmup 2 --streams 1 $2"
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
i=0
while read -r LINE; do
  [ -z "$LINE" ] && continue
  i=$((i+1))
  timeout -k 10 ${ANATOMY_TIMEOUT:-150} rocprofv3 --pmc $LINE --output-format csv -d $O/pass$i -- python3 $R/bench.py $ARGS > $O/pass$i.json 2> $O/pass$i.log
  RC=$?; echo "pass $i ($LINE): rc $RC"
  [ $RC -ge 124 ] && { echo "pass $i timed out: stopping"; break; }
done <<'PASSES'
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
GRBM_GUI_ACTIVE TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum
GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
GRBM_GUI_ACTIVE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_TC_STALL
PASSES
python3 $R/tools/pmc_anatomy.py $O ${3:-0} | tee $O/anatomy.txt
