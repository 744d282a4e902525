set -o pipefail
for s in 1 2; do ADVX_BENCH_STRIDE=$s python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stride $s', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['timed_launches'], d['roofline']['kernel_ms'], d['in_cache']['ms_per_step'], d['roofline']['traffic'])"; done
