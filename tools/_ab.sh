set -u
run() { timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['value'], d['ms_per_step'], {n[:8]:v['ms'] for n,v in k.items()}, d['config']['ms_per_step_hipgraph_replay'])"; }
for i in 1 2; do
echo "== noenv default(full)"; run --no-env
echo "== noenv SGR_FWD_MODE=half2"; SGR_FWD_MODE=half2 run --no-env
echo "== env SGR_FWD_MODE=half3"; SGR_FWD_MODE=half3 run
echo "== env SGR_FWD_MODE=half2"; SGR_FWD_MODE=half2 run
done
