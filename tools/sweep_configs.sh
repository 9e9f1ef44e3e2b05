# frames/s of chosen configurations under environment variants: tools/sweep_configs.sh "C1 C2" "VAR=1" "VAR=2" ...
cfgs=$1; shift
for v in "$@"; do
  for c in $cfgs; do
    env $v timeout -k 10 300 python tools/bench_configs.py 32 $c | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('$v', r['config'][:12], r['fps'], r['kernel_us_per_frame']['raster'])"
  done
done
