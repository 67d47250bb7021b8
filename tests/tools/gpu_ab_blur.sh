# blurred 1024x2048 map through the general path with two library builds (MN_LIB), alternating
for pass in 1 2; do
for lib in base alt; do
  if [ $lib = alt ]; then export MN_LIB=$PWD/mergenet_amd/libmergenet_hip_alt.so; else unset MN_LIB; fi
  python tests/tools/gpu_blur_sweep.py 2>&1 | grep "^{}" | sed "s/^/pass $pass lib $lib: /"
done
done
