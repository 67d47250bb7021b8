python bench.py --no-cpu-baseline --no-general-path --no-pipelined 2>/dev/null | python tools/show_bench.py /dev/stdin | head -2
python bench.py --no-cpu-baseline --no-general-path --no-pipelined --mode 2 --steps 6 --warmup 1 2>/dev/null | python tools/show_bench.py /dev/stdin | head -2
MN_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | python tools/show_bench.py /dev/stdin | tail -3
