// What the memory system gives a kernel shaped like the affinity-scoring sweep, with the arithmetic taken out:
// R float planes of N values read once by 16-byte loads (a lane takes 4 consecutive pixels of every plane, as
// mn_cc_sign does), W1 4-byte and W2 1-byte values per pixel written.  Prints microseconds per launch (HIP
// events over many launches, inputs rotated through NBUF sets so that the 256 MB Infinity Cache cannot keep them)
// and the rate in algorithmic bytes (reads only, as roofline.achieved counts them) and in total bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_ceiling.hip -o tools/stream_ceiling && tools/stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int DEPTH, int THREADS>
__global__ __launch_bounds__(THREADS) void stream_like_sweep(const float* __restrict__ in, int R, size_t N,
                                                             unsigned* __restrict__ out4, int W1,
                                                             unsigned char* __restrict__ out1, int W2) {
  const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;      // group of four pixels
  if (4 * i >= N) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r0 = 0; r0 < R; r0 += DEPTH) {
    float4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
      v[d] = (r0 + d < R) ? *reinterpret_cast<const float4*>(in + (size_t)(r0 + d) * N + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { acc.x += v[d].x; acc.y = fmaxf(acc.y, v[d].y); acc.z += v[d].z; acc.w = fmaxf(acc.w, v[d].w); }
  }
  const uint4 o = make_uint4(__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w));
  for (int w = 0; w < W1; w++)                      // W1 planes of 4 bytes per pixel
    *reinterpret_cast<uint4*>(out4 + (size_t)w * N + 4 * i) = o;
  for (int w = 0; w < W2; w++)                      // W2 planes of 1 byte per pixel
    *reinterpret_cast<unsigned*>(out1 + (size_t)w * N + 4 * i) = o.x;
}

template <int DEPTH, int THREADS>
static void run(const char* what, float* in, int nbuf, int R, size_t N, unsigned* out4, int W1, unsigned char* out1, int W2) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int launches = 400;
  const unsigned grid = (unsigned)((N / 4 + THREADS - 1) / THREADS);
  for (int it = 0; it < 20; it++)
    hipLaunchKernelGGL((stream_like_sweep<DEPTH, THREADS>), dim3(grid), dim3(THREADS), 0, 0, in + (size_t)(it % nbuf) * R * N, R, N, out4, W1, out1, W2);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int it = 0; it < launches; it++)
    hipLaunchKernelGGL((stream_like_sweep<DEPTH, THREADS>), dim3(grid), dim3(THREADS), 0, 0, in + (size_t)(it % nbuf) * R * N, R, N, out4, W1, out1, W2);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / launches;
  const double rd = 4.0 * R * N, wr = (4.0 * W1 + W2) * N;
  printf("%-44s depth %d threads %3d: %7.2f us per launch  reads %6.1f MB -> %5.2f TB/s  reads+writes %6.1f MB -> %5.2f TB/s\n",
         what, DEPTH, THREADS, us, rd / 1e6, rd / us / 1e6, (rd + wr) / 1e6, (rd + wr) / us / 1e6);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main(int argc, char** argv) {
  const int H = argc > 2 ? atoi(argv[1]) : 1024, Wd = argc > 2 ? atoi(argv[2]) : 2048;
  const int R = argc > 3 ? atoi(argv[3]) : 19;
  const size_t N = (size_t)H * Wd;
  const int nbuf = 4;                                 // 4 x 159 MB of inputs: beyond the Infinity Cache
  float* in; unsigned* out4; unsigned char* out1;
  CK(hipMalloc(&in, (size_t)nbuf * R * N * 4));
  CK(hipMalloc(&out4, (size_t)8 * N * 4));
  CK(hipMalloc(&out1, (size_t)16 * N));
  CK(hipMemset(in, 0x3c, (size_t)nbuf * R * N * 4));
  printf("%d x %d, %d planes read\n", H, Wd, R);
  // reads only
  run<1, 256>("reads only", in, nbuf, R, N, out4, 0, out1, 0);
  run<5, 256>("reads only", in, nbuf, R, N, out4, 0, out1, 0);
  run<10, 256>("reads only", in, nbuf, R, N, out4, 0, out1, 0);
  run<5, 512>("reads only", in, nbuf, R, N, out4, 0, out1, 0);
  run<5, 128>("reads only", in, nbuf, R, N, out4, 0, out1, 0);
  // the sweep's writes: edge masks 4 B + class log-products 9 B + negative edges ~4 B + cls0 1 B per pixel
  run<5, 256>("reads + 17 B/px written (4 x 4 B + 1 B)", in, nbuf, R, N, out4, 4, out1, 1);
  run<10, 256>("reads + 17 B/px written (4 x 4 B + 1 B)", in, nbuf, R, N, out4, 4, out1, 1);
  run<5, 256>("reads + 5 B/px written (4 B + 1 B)", in, nbuf, R, N, out4, 1, out1, 1);
  // same inputs every launch (what a bench loop over one image sees: the Infinity Cache may hold part)
  run<5, 256>("reads only, one input set", in, 1, R, N, out4, 0, out1, 0);
  run<5, 256>("reads + 17 B/px, one input set", in, 1, R, N, out4, 4, out1, 1);
  return 0;
}
