#!/bin/bash
# frames/s and strip-kernel time per frame against strip height, batch size and lanes: tools/strip_matrix.sh [frames per step]
F=${1:-32}
for lanes in 1 4; do for b in 1 2 4 8 16 32; do for s in 1 2 4; do
  v=$(MI355R_LANES=$lanes MI355R_BATCH=$b MI355R_STRIP=$s timeout -k 10 100 python3 bench.py --steps 3 --warmup 1 --frames $F --no-cpu --no-other | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), round(r['kernel_ms_per_frame']['raster']*1000,1))") || exit 1
  echo "lanes $lanes batch $b strip $s: $v"
done; done; done
