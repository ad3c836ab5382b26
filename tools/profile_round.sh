#!/bin/bash
# Collects the round's evidence on the GPU box: PMC traffic first (bench.py copies it into roofline.traffic), then the
# bench lines and the rocprofv3 kernel stats of the same commands.  Copy what should be judged into profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for WL in cornell mesh1m; do
  rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $O/pmc_fetch_$WL -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --workload $WL > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $O/pmc_write_$WL -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --workload $WL > /dev/null 2>&1
done
python3 $R/tools/pmc_traffic.py $O $O/pmc_traffic.json > /dev/null && cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cornell -- python3 $R/bench.py --no-cpu-baseline > $O/stats_cornell.json 2> $O/stats_cornell.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mesh1m -- python3 $R/bench.py --no-cpu-baseline --workload mesh1m > $O/stats_mesh1m.json 2> $O/stats_mesh1m.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mesh1m_d4 -- python3 $R/bench.py --no-cpu-baseline --workload mesh1m --depth 4 --steps 20 > $O/stats_mesh1m_d4.json 2> $O/stats_mesh1m_d4.log
cd $R
python bench.py > $O/bench_cornell.json 2> $O/bench_cornell.log
python bench.py --workload mesh1m > $O/bench_mesh1m.json 2> $O/bench_mesh1m.log
python bench.py --workload mesh1m --depth 4 --steps 20 > $O/bench_mesh1m_d4.json 2> $O/bench_mesh1m_d4.log
cat $O/bench_cornell.json $O/bench_mesh1m.json $O/bench_mesh1m_d4.json
find $O -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -8 $f; done
