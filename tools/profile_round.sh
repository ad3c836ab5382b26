#!/bin/bash
# Collects the round's evidence on the GPU box: PMC passes first (bench.py copies their summary into roofline.traffic /
# roofline.valu_issue), then the bench lines and the rocprofv3 kernel stats of the same commands, the builders, the
# per-block lane utilisation and the VALU issue-rate microbenchmark.  Copy what should be judged into profiles/.
# usage (on the box): bash tools/profile_round.sh rNN
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for CFG in "cornell 1" "mesh1m 1" "mesh1m 4"; do
  set -- $CFG; WL=$1; D=$2; ARGS="--no-cpu-baseline --steps 5 --warmup 2 --workload $WL --depth $D --spp 1"
  rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $O/pmc_fetch_${WL}_d$D -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $O/pmc_write_${WL}_d$D -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_${WL}_d$D -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d $O/pmc_grbm_${WL}_d$D -- python3 $R/bench.py $ARGS > /dev/null 2>&1
  echo "pmc $WL d$D done"
done
python3 $R/tools/pmc_traffic.py $O $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1 && cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
for CFG in "cornell 1" "mesh1m 1" "mesh1m 4"; do
  set -- $CFG; WL=$1; D=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${WL}_d$D -- python3 $R/bench.py --no-cpu-baseline --workload $WL --depth $D --spp 1 --steps 100 > $O/stats_${WL}_d$D.json 2> $O/stats_${WL}_d$D.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_builders -- python3 $R/tools/build_times.py > $O/build_times.txt 2>&1
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --workload mesh1m --resolution 3840x2160 > $O/bench_config5_world1.json 2> $O/bench_config5_world1.log
python tools/builder_quality.py sbvh sah lbvh ploc4 ploc16 ploc64 > $O/builder_quality.txt 2>&1
for A in "mesh1m 1" "mesh1m 4" "cornell 1"; do python tools/lane_util.py $A; done > $O/lane_util.txt 2>&1
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate > $O/valu_rate.txt 2>&1
cat $O/bench_default.json
find $O -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -6 $f; done
cat $O/pmc_traffic.json
