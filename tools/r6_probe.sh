#!/bin/bash
# round-6 probe: fingerprints of the library in place, headline timing samples (default / agent-scope acquire), stamps of k_start and of the workers
O=${1:-gpurun_out/r6probe}; TAG=${2:-base}; mkdir -p $O
(timeout 600 python tools/bits_dump.py $TAG) > $O/bits_$TAG.txt 2>&1
(for i in 1 2 3; do timeout 200 python tools/ab_time.py 4096 zamlf_n30_nx6 $TAG; done
 timeout 200 python tools/ab_time.py 256 zamlf_n30_nx6 $TAG
 timeout 200 python tools/ab_time.py 4096 usalf_n50_nx5 $TAG
 timeout 200 python tools/ab_time.py 1024 ca $TAG
 timeout 200 python tools/start_timing.py 4096
 timeout 200 python tools/pipe_timing.py 4096 0
 timeout 200 python tools/res_timing.py 256 1 -1) 2>&1 | grep -v "amdgpu.ids" > $O/perf_$TAG.txt
cat $O/bits_$TAG.txt $O/perf_$TAG.txt
