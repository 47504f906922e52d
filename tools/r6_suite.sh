#!/bin/bash
# round-6 check of the library in place: GPU suite, hand-off stress, kernel timeline of the headline solve under rocprofv3
O=${1:-gpurun_out/r6suite}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/tests.log
(timeout 600 python tools/pipe_stress.py ${2:-1500} 2>&1 | grep -v amdgpu.ids | tail -12) > $O/stress.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/head -o head --output-format csv -- python bench.py --headline-only --no-cpu-baseline --no-traffic > $O/head.log 2>&1
python tools/solve_timeline.py $O/head/head_kernel_trace.csv > $O/solve_timeline.txt 2>&1
bash tools/r6_quick.sh suite > $O/quick.log 2>&1
cat $O/tests.log $O/stress.log $O/solve_timeline.txt $O/quick.log; tail -1 $O/head.log | cut -c1-600
