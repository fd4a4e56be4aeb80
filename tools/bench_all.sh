set -x
python bench.py > gpurun_out/bench_1024.json 2> gpurun_out/bench_1024.err; tail -c 3000 gpurun_out/bench_1024.json
python bench.py --workload pond --steps 3200 > gpurun_out/bench_pond.json 2>&1; tail -c 1500 gpurun_out/bench_pond.json
python bench.py --workload renderer1024 --steps 2000 > gpurun_out/bench_renderer.json 2>&1; tail -c 1200 gpurun_out/bench_renderer.json
python bench.py --workload ocean4096 --steps 128 --warmup 32 > gpurun_out/bench_4096.json 2>&1; tail -c 2500 gpurun_out/bench_4096.json
MW_BENCH_BACKEND=gloo MW_BENCH_SAME_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 640 --warmup 64 > gpurun_out/bench_2rank.json 2>&1; tail -c 900 gpurun_out/bench_2rank.json
