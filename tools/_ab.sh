timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-900
