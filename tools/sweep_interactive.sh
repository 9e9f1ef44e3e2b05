# the frame-at-a-time loop (tools/interactive.py) under each of the environment settings given, twice over:
# tools/sweep_interactive.sh "<env 1>" "<env 2>" ...
for round in 1 2; do
  for v in "$@"; do
    echo "== $v"
    env $v timeout -k 10 120 python3 tools/interactive.py helmet 400 2>/dev/null | tail -1 || exit 1
  done
done
