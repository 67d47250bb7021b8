for b in 1 8; do
MN_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 203 --warmup 5 --no-cpu-baseline --spin-seconds 0.5 --exchange-batch $b 2>/dev/null | python tools/show_bench.py /dev/stdin | grep "value\|distributed\|id_match"
done
python -m pytest tests/test_gpu_prepare.py -q -m gpu 2>&1 | tail -2
