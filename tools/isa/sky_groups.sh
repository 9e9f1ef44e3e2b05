#!/bin/bash
# VALU instructions per skybox fragment, by group (tools/isa/sky_groups.hip), compiled like the library
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -Wno-unused-function -S --cuda-device-only "$HERE/sky_groups.hip" -o /tmp/sky_groups.s
base=0
for k in g0_empty g1_weights g2_perspective g3_varyings g4_cube_face g5_texel_address g6_fetch_decode g7_saturate_pack g8_whole_fragment g9_bytes_fragment; do
  awk -v k="$k" '$0 ~ "^_Z[0-9]*"k {f=1} f{print} f && /s_endpgm/{exit}' /tmp/sky_groups.s > /tmp/grp.s
  valu=$(grep -c '^\s*v_' /tmp/grp.s || true); salu=$(grep -c '^\s*s_' /tmp/grp.s || true); mem=$(grep -c '^\s*\(global\|buffer\|flat\)_' /tmp/grp.s || true)
  div=$(grep -c 'v_div_scale' /tmp/grp.s || true)
  [ "$k" = g0_empty ] && base=$valu
  echo "$k: VALU $valu (net of the empty kernel's $base: $((valu - base))), SALU $salu, memory $mem, div_scale $div"
done
