# texture layouts under the pbr shade kernel: 8x8 row-major tiles (0), 16x16 tiles (1), 8x8 tiles in Z order (2)
mkdir -p gpurun_out
export TMPDIR=/tmp
for layout in 0 1 2; do
  export MI355R_TEX_LAYOUT=$layout
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-other > gpurun_out/layout_$layout.json || exit 1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/layout_pmc_$layout/fetch -- python3 bench.py --steps 3 --warmup 1 --frames 32 --no-cpu --no-other > gpurun_out/layout_pmc_$layout.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/layout_pmc_$layout/hit -- python3 bench.py --steps 3 --warmup 1 --frames 32 --no-cpu --no-other >> gpurun_out/layout_pmc_$layout.log 2>&1 || exit 1
  python3 - $layout <<'PY'
import csv, glob, json, sys
from collections import defaultdict
layout = sys.argv[1]
r = json.load(open(f'gpurun_out/layout_{layout}.json'))
pmc = defaultdict(list)
for f in glob.glob(f'gpurun_out/layout_pmc_{layout}/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'rz_shade_list_kernel<1' in row['Kernel_Name']:
            pmc[row['Counter_Name']].append(float(row['Counter_Value']))
med = {k: sorted(v)[len(v) // 2] for k, v in pmc.items()}
hit, miss = med.get('TCC_HIT_sum', 0), med.get('TCC_MISS_sum', 1)
print(f"layout {layout}: {r['value']:.0f} fps, pbr shade {r['kernel_ms_per_frame']['shade_pbr'] * 1e3:.2f} us/frame, "
      f"L2 hit rate {hit / (hit + miss):.3f}, FETCH_SIZE {med.get('FETCH_SIZE', 0) * 1024 / 1e6:.0f} MB/launch (raw)")
PY
done
