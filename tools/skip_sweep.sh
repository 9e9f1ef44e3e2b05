for k in 5 6 7; do
  export MI355R_DEBUG_SKIP=$k
  echo "skip=$k"
  bash tools/trace.sh skip$k | grep "rz_bin" || exit 1
done
