#!/bin/bash
# stall anatomy of the headline kernel and of the four-segment frame
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5x; mkdir -p $O
bash $R/tools/pmc_anatomy.sh $O/d1 "--workload mesh1m --depth 1 --spp 4" > $O/d1.log 2>&1; tail -5 $O/d1.log
bash $R/tools/pmc_anatomy.sh $O/d4 "--workload mesh1m --depth 4 --spp 4" > $O/d4.log 2>&1; tail -5 $O/d4.log
