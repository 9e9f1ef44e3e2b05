#!/bin/bash
# instructions per pbr fragment, by group (tools/isa/pbr_groups.hip), compiled like the library
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -Wno-unused-function -S --cuda-device-only "$HERE/pbr_groups.hip" -o /tmp/pbr_groups.s
for k in p0_empty p1_varyings p2_surface p3_ambient p4_light p5_display p6_whole; do
  awk -v k="$k" '$0 ~ "^_Z[0-9]*"k {f=1} f{print} f && /^\.Lfunc_end/{exit}' /tmp/pbr_groups.s > /tmp/grp_$k.s
  valu=$(grep -c '^\s*v_' /tmp/grp_$k.s || true); f64=$(grep -c '^\s*v_[a-z_0-9]*f64' /tmp/grp_$k.s || true); salu=$(grep -c '^\s*s_' /tmp/grp_$k.s || true)
  mem=$(grep -c '^\s*\(global\|buffer\|flat\|scratch\)_' /tmp/grp_$k.s || true); div=$(grep -c 'v_div_scale' /tmp/grp_$k.s || true)
  sq=$(grep -c 'v_sqrt\|v_rsq' /tmp/grp_$k.s || true)
  echo "$k: VALU $valu (f64 $f64), SALU $salu, memory $mem, div_scale $div, sqrt $sq"
done
