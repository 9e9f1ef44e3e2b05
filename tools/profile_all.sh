#!/bin/bash
# tools/profile.sh for several configurations, results gathered under gpurun_out/profiles_r03/ in the names
# profiles/ keeps them under: r03_<config>_kernel_stats.csv, r03_<config>_summary.txt, traffic_<config>.json
# usage: tools/profile_all.sh C1 C3 ...
set -u
DEST=gpurun_out/profiles_r03
mkdir -p "$DEST"
for C in "$@"; do
    bash tools/profile.sh "r03_$C" "$C" > "$DEST/r03_${C}_profile.log" 2>&1 || { tail -5 "$DEST/r03_${C}_profile.log"; exit 1; }
    cp "gpurun_out/prof_r03_$C/summary.txt" "$DEST/r03_${C}_summary.txt"
    cp "gpurun_out/prof_r03_$C/traffic.json" "$DEST/traffic_$C.json"
    cp $(ls gpurun_out/prof_r03_$C/trace/*/*kernel_stats.csv | head -1) "$DEST/r03_${C}_kernel_stats.csv"
    echo "$C done"
done
