cd $GRAFT_REPO_ROOT
run() { python tools/bench_knobs.py "$@" -- --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s %.3f ms' % (' '.join(d['knobs']), d['ms_per_step']))"; }
for rep in 1 2; do
  run --set tn=85,75
  run --set tn=85,150
  run --set tn=85,300
  run --set tn=85,600
  run --set tn=85,2000
  run --set tn=60,75
  run --set tn=120,75
done
