// mn_ref_logf.h -- glibc's logf, bit for bit, for device AND host (the host build is what
// tests/test_ref_logf.py compiles with gcc and compares with the C library's logf).
//
// The reference calls log(float) = logf for the class and sameness terms (utils/csegment/segment.h:296,
// segment.cc:35); its float32 merge decisions depend on the exact values, and a correctly rounded logf
// differs from glibc's in ~0.1 % of the inputs.  glibc 2.35 (the image's libm; the published algorithm
// of sysdeps/ieee754/flt-32/e_logf.c, from ARM's optimized routines) evaluates, in double precision,
//   log x = log1p(z / c - 1) + log c + k ln 2,  16-entry table of (1/c, log c), cubic in r = z/c - 1.
// Restated here with the operation order of the FMA variant x86-64 dispatches to (the SSE2 variant gives
// the same float on every input); checked against the host's logf on all 2.13e9 positive normal floats
// when it was written, and on a sample of them by the CPU suite.  Inputs of the merger are clipped to
// [2^-23, 1 - 2^-23] (c_segment.pyx:53-55): zero, subnormals, infinities and NaN do not occur and are
// not handled.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define MN_REF_HD __host__ __device__ __forceinline__
#else
#define MN_REF_HD static inline
#endif

MN_REF_HD float mn_ref_logf(float x) {
  const double T[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
      {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
      {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
      {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
      {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
      {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
      {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
  unsigned ix;
  __builtin_memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0.0f;
  const unsigned tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const int k = (int)tmp >> 23;
  const unsigned iz = ix - (tmp & 0xff800000u);
  float zf;
  __builtin_memcpy(&zf, &iz, 4);
  const double z = (double)zf;
  const double invc = T[i][0], logc = T[i][1];
  const double y0 = __builtin_fma((double)k, 0x1.62e42fefa39efp-1, logc);
  const double r = __builtin_fma(z, invc, -1.0);
  double y = __builtin_fma(r, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2);
  const double r2 = r * r;
  const double t = r + y0;
  y = __builtin_fma(r2, -0x1.00ea348b88334p-2, y);
  return (float)__builtin_fma(r2, y, t);
}

// glibc's expf, bit for bit (the reference's same_different_bias path: `exp(-logit)` on a float is expf,
// utils/csegment/segment.cc:183-195).  glibc 2.35's published algorithm (sysdeps/ieee754/flt-32/e_expf.c, from
// ARM's optimized routines): in double precision, x * 32 / ln 2 = k + r with k rounded to nearest (adding and
// subtracting 1.5 * 2^52), exp x = 2^(k/32) * (C0 r^3 + C1 r^2 + C2 r + 1), 2^(i/32) from a 32-entry table whose
// exponent field takes k's upper bits.  Operation order of the FMA variant x86-64 dispatches to.  |x| < 88 only
// (the logit of a clipped probability plus a bias: |x| <= 16 + |bias|); overflow, underflow and NaN are not
// handled.  Equal to the C library's expf on all 6.1e8 floats with 2^-30 <= |x| < 80 when it was written;
// tests/test_ref_logf.py compares the host build with the C library's expf on a sample.
MN_REF_HD float mn_ref_expf(float x) {
  const unsigned long long T[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
      0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
      0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
      0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
      0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  const double shift = 0x1.8p+52;
  const double xd = (double)x;
  const double z = (0x1.71547652b82fep+0 * 32.0) * xd;
  double kd = z + shift;
  unsigned long long ki;
  __builtin_memcpy(&ki, &kd, 8);
  kd -= shift;
  const double r = __builtin_fma(0x1.71547652b82fep+0 * 32.0, xd, -kd);   // (the FMA variant fuses z - kd with z's product)
  unsigned long long t = T[ki & 31ull];
  t += ki << (52 - 5);
  double s;
  __builtin_memcpy(&s, &t, 8);
  const double c0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, c1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0,
               c2 = 0x1.62e42ff0c52d6p-1 / 32.0;
  const double zz = __builtin_fma(c0, r, c1);
  const double r2 = r * r;
  double y = __builtin_fma(c2, r, 1.0);
  y = __builtin_fma(zz, r2, y);
  y = y * s;
  return (float)y;
}
