# longest bin list per configuration, as the library reports it (MI355R_VERBOSE)
for c in C1 C2 C3 C4 C5 S1; do
  echo "== $c"; MI355R_VERBOSE=1 timeout -k 10 300 python tools/bench_configs.py 32 $c 2>&1 | grep "longest bin list" | sort | uniq -c | sort -rn | head -4
done
