# the frame-at-a-time loop's kernels under each of the environment settings given (kernel statistics): tools/sweep_team.sh "<env 1>" ...
for v in "$@"; do
  echo "== $v"
  env $v bash tools/stats_cmd.sh 4 python3 tools/interactive.py helmet 400 2>&1 | grep -E "strip|sky|launches" || exit 1
done
