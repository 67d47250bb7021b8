for c in "2.5 20" "0 20" "2.5 20" "2.5 5"; do
    set -- $c
    python bench.py --no-cpu-baseline --no-general-path --no-pipelined --spin-seconds $1 --warmup $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('spin $1 warmup $2 %8.1f Mpixel/s  %.4f ms/step  sweep %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
done
MN_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 4 --no-cpu-baseline --spin-seconds 1 2>/dev/null | python tools/show_bench.py /dev/stdin | tail -2
