set -u
O=${1:-gpurun_out/r6prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo bench rc=$?
timeout 600 rocprofv3 --kernel-trace --stats -d $O/head -o head --output-format csv -- python bench.py --headline-only --no-cpu-baseline --no-traffic > $O/head.log 2>&1; echo head rc=$?
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o full --output-format csv -- python bench.py --no-cpu-baseline --no-traffic > $O/prof.log 2>&1; echo prof rc=$?
python tools/trace_split.py $O/prof/full_kernel_trace.csv > $O/kernel_trace_split.txt 2>&1
python tools/solve_timeline.py $O/head/head_kernel_trace.csv > $O/solve_timeline.txt 2>&1
timeout 1500 bash tools/pmc_run.sh $O/pmc > $O/pmc_run.log 2>&1; echo pmc rc=$?
python tools/pmc_summary.py $O/pmc --json $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1
python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1
( timeout 200 python tools/pipe_timing.py 4096 0; timeout 200 python tools/pipe_timing.py 4096 20; timeout 200 python tools/res_timing.py 256 1 -1 ) 2>&1 | grep "mpcgpu\|last pass" > $O/worker_stamps.txt
ls $O $O/head $O/prof | head -40
cat $O/bench.json
