#!/bin/bash
# instruction-cache behaviour of the headline kernel
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/r4aq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --gpus 1 --workload mesh1m --depth 1 --spp 4 --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --no-oracle-check --settle-ms 0 --streams 1"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE --output-format csv -d $OUT/ic -- $B > $OUT/ic.json 2> $OUT/ic.log || echo "ic failed"
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- $B > $OUT/sq.json 2> $OUT/sq.log || echo "sq failed"
python3 $R/tools/pmc_counters.py $OUT/ic $OUT/sq | grep -v STATS | grep "k_segment" | head -20
