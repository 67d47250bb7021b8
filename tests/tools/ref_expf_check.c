/* Host check of mn_ref_expf (mergenet_amd/csrc/mn_ref_logf.h) against the C library's expf (tests/test_ref_logf.py).
 * usage: ref_expf_check STRIDE  -- every STRIDE-th float of [2^-30, 64] and of [-64, -2^-30]; prints
 * "checked N mismatches M". */
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
    a = mn_ref_expf(x);
    b = expf(x);
    if (memcmp(&a, &b, 4)) bad++;
    (*n)++;
  }
  return bad;
}

int main(int argc, char** argv) {
  const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 97u;
  long n = 0, bad = 0;
  bad += check(0x30800000u, 0x42800000u, stride, &n);   /* 2^-30 .. 64 */
  bad += check(0xb0800000u, 0xc2800000u, stride, &n);   /* -2^-30 .. -64 */
  bad += check(0x41000000u, 0x41900000u, 1, &n);        /* [8, 18]: the logits of clipped probabilities */
  bad += check(0xc1000000u, 0xc1900000u, 1, &n);
  printf("checked %ld mismatches %ld\n", n, bad);
  return bad != 0;
}
