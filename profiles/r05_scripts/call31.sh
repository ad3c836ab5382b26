# single-stream kernel trace of the four-segment frame on the current build (durations, gaps)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5af; mkdir -p $O
for S in 1 2; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s$S -- python3 $R/bench.py --no-cpu-baseline --no-live-pmc --no-oracle-check --workload mesh1m --depth 4 --spp 4 --steps 20 --warmup 5 --streams $S > $O/trace_s$S.json 2> $O/trace_s$S.log
done
python3 - <<'PY'
import csv,glob,os
for S in (1,2):
    f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+f'/gpurun_out/r5af/trace_s{S}/**/*kernel_trace.csv',recursive=True)[0]
    rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
    seq=[(r['Kernel_Name'].split('(')[0][:58],int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Queue_Id']) for r in rows]
    mid=len(seq)*2//3
    print('streams',S)
    t0=seq[mid][1]
    for k,s,e,q in seq[mid:mid+(14 if S==1 else 26)]:
        print(f"  {k:60s} start {(s-t0)/1e3:9.1f} dur {(e-s)/1e3:8.1f} us q {q}")
PY
