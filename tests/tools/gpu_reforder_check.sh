# The reference-order loop after a change: small vectors, its tests, then the larger vectors (minutes).
MN_TRACE_EXACT=1 timeout -k 10 120 python tests/tools/gpu_reforder.py cseg_adv_32x32_o0 cseg_synth_32x64_n60 cseg_synth_48x80_c81 cseg_blur4_128x256_s5100 > gpurun_out/reforder_small.log 2>&1
grep "tie order 1" gpurun_out/reforder_small.log
timeout -k 10 200 python -m pytest tests/test_gpu_exact.py -x -q -m gpu -k "reference_order or tie" > gpurun_out/reforder_tests.log 2>&1
tail -3 gpurun_out/reforder_tests.log
if [ "$1" = "big" ]; then
  MN_TRACE_EXACT=1 timeout -k 10 700 python tests/tools/gpu_reforder.py cseg_blur_256x512_r2 cseg_crowd48_256x512_s6408 cseg_synth_400x667_c81 cseg_synth_512x1024_s1000 > gpurun_out/reforder_big.log 2>&1
  grep "tie order" gpurun_out/reforder_big.log | cut -c1-200
fi
