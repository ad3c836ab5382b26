#!/bin/bash
# Collects the round's evidence on the GPU box.  Copy what should be judged into profiles/ (tools/profile_collect.py does).
#   1. rocprofv3 --kernel-trace --stats of the bench's own headline command (top-level block only): the kernel's average duration there
#      is what roofline.launch_ms must agree with
#   2. the --pmc passes per workload (three separate passes: FETCH_SIZE + GRBM, WRITE_SIZE, SQ), summarised by tools/pmc_traffic.py
#   3. the default bench line, the one-process multi-device rehearsal, lane utilisation, the VALU issue microbenchmark
# usage (on the box): bash tools/profile_round.sh rNN
TAG=${1:-r05}
# PART=1: kernel stats + counter passes; PART=2: bench lines, probes; unset: everything (may not fit one 20-minute gpurun call)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round_$TAG; [ "$PART" != "2" ] && rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != "2" ]; then
HEAD="--no-cpu-baseline --no-live-pmc --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -- python3 $R/bench.py $HEAD > $O/stats_headline.json 2> $O/stats_headline.log
echo "headline stats done"
for CFG in "mesh1m 4 4" "cornell 1 1"; do
  set -- $CFG; WL=$1; D=$2; S=$3
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${WL}_d$D -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --workload $WL --depth $D --spp $S > $O/stats_${WL}_d$D.json 2> $O/stats_${WL}_d$D.log
done
echo "kernel stats done"
for CFG in "mesh1m 1 4" "mesh1m 4 4" "cornell 1 1" "mesh520 1 4 --device-built sah" "mesh520 4 4 --device-built sah"; do
  set -- $CFG; WL=$1; D=$2; S=$3; shift 3
  ARGS="--no-cpu-baseline --no-live-pmc --settle-ms 0 --steps 5 --warmup 2 --streams 1 --workload $WL --depth $D --spp $S $*"
  rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d $O/pmc_fetch_${WL}_d$D -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $O/pmc_write_${WL}_d$D -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_${WL}_d$D -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  echo "pmc $WL d$D done"
done
python3 $R/tools/pmc_traffic.py $O $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
fi
[ "$PART" = "1" ] && exit 0
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.log
echo "bench default done"
python bench.py --gpus 1 --one-process --virtual-devices 8 --no-cpu-baseline --no-live-pmc > $O/bench_one_process_virtual8.json 2> $O/bench_one_process_virtual8.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --workload mesh1m --resolution 3840x2160 --no-cpu-baseline > $O/bench_config5_world1.json 2> $O/bench_config5_world1.log
for A in "mesh1m 1" "mesh1m 1 lanes" "mesh1m 4" "mesh1m 4 lanes" "mesh1m 4 inplace_shadow=1" "mesh1m 4 lanes_per_ray=1" "cornell 1"; do python tools/lane_util.py $A; done > $O/lane_util.txt 2>&1
for A in "mesh1m 1" "mesh1m 4"; do python tools/lane_hist.py $A; done > $O/lane_hist.txt 2>&1
python tools/build_probe.py > $O/build_probe.txt 2>&1
hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_issue_cycles tools/ubench/valu_issue_cycles.hip > $O/valu_issue_cycles.build.log 2>&1 && /tmp/valu_issue_cycles > $O/valu_issue_cycles.txt 2>&1
python tools/shard_times.py 3840x2160 1 4 > $O/shard_times_4k.txt 2>&1
python tools/roofline.py frac $O/bench_default.json > $O/roofline_frac.txt 2>&1
cat $O/bench_default.json | head -c 3000; echo
find $O -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -5 $f; done
