#!/bin/bash
# kernel trace of 128-frame steps of one scene and how busy the GPU was: tools/busy.sh helmet|buster|...
# (gpurun merges the trace into gpurun_out/busy_<scene>/ beside older ones: read the newest *kernel_trace.csv)
export TMPDIR=/tmp
OUT=gpurun_out/busy_$1
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/host_cost.py $1 - 1920 1080 128 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
grep -E "submit|record" $OUT/log.txt
python3 tools/busy_share.py $OUT
