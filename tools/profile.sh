#!/bin/bash
# Profiles one workload on the GPU box: a kernel-trace/stats pass, then PMC passes (each counter group in
# a run of its own; never combined with tracing).
# usage: tools/profile.sh <tag> [C1|C2|C3|C4|C5]   -> gpurun_out/prof_<tag>/ (summary.txt, traffic.json)
# C2 (default) = bench.py's own loop on the helmet; the others = tools/bench_configs.py on that configuration.
set -u
TAG=${1:-run}
CONFIG=${2:-C2}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
# one kernel at a time: a batch's sky kernel in line, not beside its shade kernels (counters of kernels that run side by side
# cannot be told apart, and their durations stretch)
export MI355R_SKY_SIDE=0
# ... and crowded bins in their strips, not in rz_crowd_kernel beside them (azura: 3 % slower than the default)
export MI355R_CROWD_MIN=0
if [ "$CONFIG" = C2 ]; then
    BENCH="python3 bench.py --steps 3 --warmup 1 --frames 32 --no-cpu --no-other"
else
    BENCH="python3 tools/bench_configs.py 32 $CONFIG plain"
fi
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
pass() {
    name=$1; shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass td TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE
python3 tools/summarize_profile.py "$OUT" 32 > "$OUT/summary.txt" 2>&1
grep -v "^    " "$OUT/summary.txt" | head -24
