# same library, one environment variable on / off (GPU box): bash tools/ab_env.sh VAR <bench_configs names / c4 / c5>
R=$GRAFT_REPO_ROOT; VAR=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    echo "== $cfg  with $VAR=1"; env $VAR=1 python3 $R/bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
    echo "== $cfg  without";       python3 $R/bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
