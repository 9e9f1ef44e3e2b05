# the fuzzer under the library's alternative code paths
for v in "MI355R_STRIP_LONG=1" "MI355R_STRIP_LONG=300" "MI355R_SKY_TABLE=0" "MI355R_SKY_SIDE=0" "MI355R_FILL_WAVES=4" "MI355R_BUNDLE=0" "MI355R_SKY_BYTES=0" "MI355R_PACK=0" "MI355R_STRIP=1" "MI355R_LANES=1 MI355R_BATCH=1" "MI355R_MICRO_MIN_LIST=0" "FUZZ_PER_TRIANGLE=1" "FUZZ_PER_TRIANGLE=1 MI355R_REPLAY=0"; do
  echo "== $v"
  env $v timeout -k 10 400 python -u tests/fuzz_parity.py ${1:-150} ${2:-31} 700 2>&1 | tail -1 || exit 1
done
