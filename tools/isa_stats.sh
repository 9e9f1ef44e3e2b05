#!/bin/bash
# Instruction mix of one kernel of the product library: tools/isa_stats.sh <mangled-name-substring>
# (compiles api.hip to gfx950 assembly with the library's own flags)
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -S --cuda-device-only "$HERE/renderer_amd/csrc/api.hip" -o /tmp/api.s 2>/dev/null
for k in "$@"; do
  awk -v k="$k" '$0 ~ "^_Z[A-Za-z0-9_]*"k"[A-Za-z0-9_]*:" {f=1} f{print} f && /^\.Lfunc_end/{exit}' /tmp/api.s > /tmp/kernel_$k.s
  echo "== $k: $(wc -l < /tmp/kernel_$k.s) lines, VALU $(grep -c '^\s*v_' /tmp/kernel_$k.s), f64 $(grep -c '^\s*v_[a-z_0-9]*f64' /tmp/kernel_$k.s), SALU $(grep -c '^\s*s_' /tmp/kernel_$k.s), vmem $(grep -c '^\s*\(global\|buffer\|flat\)_' /tmp/kernel_$k.s), lds $(grep -c '^\s*ds_' /tmp/kernel_$k.s)"
done
