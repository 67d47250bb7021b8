/* Host check of mergenet_amd/csrc/mn_ref_logf.h against the C library's logf (tests/test_ref_logf.py).
 * usage: ref_logf_check STRIDE  -- every STRIDE-th float of [2^-24, 2], plus whole binades around the
 * clip limits 2^-23 and 1 - 2^-23; prints "checked N mismatches M". */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../mergenet_amd/csrc/mn_ref_logf.h"

static long check(uint32_t lo, uint32_t hi, uint32_t stride, long* n) {
  long bad = 0;
  for (uint32_t ix = lo; ix < hi; ix += stride) {
    float x, a, b;
    memcpy(&x, &ix, 4);
    a = mn_ref_logf(x);
    b = logf(x);
    if (memcmp(&a, &b, 4)) bad++;
    (*n)++;
  }
  return bad;
}

int main(int argc, char** argv) {
  const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 97u;
  long n = 0, bad = 0;
  bad += check(0x33800000u, 0x40000000u, stride, &n);   /* 2^-24 .. 2 */
  bad += check(0x34000000u, 0x34800000u, 1, &n);        /* the binade of 2^-23 */
  bad += check(0x3f000000u, 0x3f800001u, 1, &n);        /* [0.5, 1] */
  printf("checked %ld mismatches %ld\n", n, bad);
  return bad != 0;
}
