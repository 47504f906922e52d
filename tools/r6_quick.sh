#!/bin/bash
# quick timing samples of the library in place
TAG=${1:-x}
(for i in 1 2 3; do timeout 200 python tools/ab_time.py 4096 zamlf_n30_nx6 $TAG; done
 timeout 200 python tools/ab_time.py 256 zamlf_n30_nx6 $TAG
 timeout 200 python tools/ab_time.py 4096 usalf_n50_nx5 $TAG
 timeout 200 python tools/ab_time.py 1024 ca $TAG) 2>&1 | grep -v "amdgpu.ids"
