for k in 3 10 4 9 8; do
  export MI355R_DEBUG_SKIP=$k
  echo "skip=$k"
  bash tools/trace.sh skip$k | grep "rz_raster" || exit 1
done
