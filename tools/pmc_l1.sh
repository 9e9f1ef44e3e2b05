# vector L1 (TCP) and texture-address (TA) counters of the shade kernels, one pass per counter group
set -u
OUT=gpurun_out/pmc_l1
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --frames 32 --no-cpu --no-other"
pass() {
    name=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum
pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass td TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    if not any(x in k for x in ('shade_list', 'sky_kernel', 'strip_kernel')): continue
    print(k)
    for name, vals in sorted(c.items()):
        vals = sorted(vals); print(f"    {name:45s} median {vals[len(vals)//2]:16.0f}  n {len(vals)}")
PY
