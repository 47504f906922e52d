#!/bin/bash
# the round-trip used while tuning: GPU suite (stops at the first failure), headline / B = 256 / N = 50 / collision-avoidance timing samples, stamps of a straggler round
O=${1:-gpurun_out/quick}; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > $O/tests.log
(for i in 1 2 3; do timeout 200 python tools/ab_time.py 4096 zamlf_n30_nx6 head; done
 timeout 200 python tools/ab_time.py 256 zamlf_n30_nx6 b256
 timeout 200 python tools/ab_time.py 4096 usalf_n50_nx5 n50
 timeout 200 python tools/ab_time.py 1024 ca ca
 timeout 200 python tools/res_timing.py 256 1 -1) 2>&1 | grep -v "amdgpu.ids" > $O/perf.log
cat $O/tests.log $O/perf.log
