# two shards on two streams, the second one a fixed phase behind the first (its step starts when the first shard's segment k has run): do an issue-bound
# first segment and memory-path-bound bounce segments share the chip better than two of a kind?
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5az; mkdir -p $O
M="--workload mesh1m --spp 4"
export AB_LIBS="default"
export AB_CHECK=" "
export AB_STEPS=50
R=""
for K in 0 1 2 3 4 0; do R="$R;d4_k$K|$M --depth 4 --option stream_skew=$K"; done
for K in 0 1 2; do R="$R;d2_k$K|$M --depth 2 --option stream_skew=$K"; done
for K in 0 2; do R="$R;hbm_k$K|--workload mesh520 --depth 4 --spp 4 --device-built sah --option stream_skew=$K"; done
export AB_RUNS="${R#;}"
bash tools/ab_run.sh $O
