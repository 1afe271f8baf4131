cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "DPIPE_STORE_FIRST=0" "DPIPE_X=1 --torch-adamw" "DPIPE_X=1 --no-graph" "DPIPE_X=1 --steps-in-flight 1"; do
  env ${v%% *} timeout 300 python bench.py --stack 2 --lanes 2 --steps 6 --warmup 2 --no-cpu-baseline --no-synced-loop $(echo "$v" | cut -s -d' ' -f2-) 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', 'value', d['value'], 'loss', d['loss'], 'gn', d['grad_norm'])"
done
