# instruction counters of the rasteriser under MI355R_STRIP variants: tools/pmc_strip.sh "<env assignments>" ...
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_strip_$i
  env $v timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_strip_$i -- python3 bench.py --steps 3 --warmup 1 --frames 32 --no-cpu --no-other > gpurun_out/pmc_strip_$i.log 2>&1 || { tail -3 gpurun_out/pmc_strip_$i.log; exit 1; }
  python3 - "$v" gpurun_out/pmc_strip_$i <<'PY'
import csv, glob, sys
from collections import defaultdict
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[2] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if k.startswith(('rz_strip', 'rz_visibility', 'rz_sky', 'rz_shade_list_kernel<1')):
            pmc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in pmc.items():
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    print(sys.argv[1], k, ' '.join(f"{n[3:]}={v / 1e6:.1f}M" for n, v in sorted(med.items())))
PY
done
