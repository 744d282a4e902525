set -o pipefail
for sc in weak strong; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --scaling $sc 2>gpurun_out/bench2.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sc', d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], d['config']['prompts_per_gpu'], d['config']['exchange'][:60], d['config']['replicas_identical'], d['config']['exchange_timed_out'], d['roofline']['frac'], d.get('in_cache',{}).get('ms_per_step'))" || tail -20 gpurun_out/bench2.err
done
