// mn_device.h -- device-side arithmetic shared by every kernel of libmergenet_hip.so.
//
// Written for gfx950 only (wave64).  All record arithmetic that decides a merge is float32
// in the reference's operation order (utils/csegment/segment.cc:107-150) and this file is
// compiled with -ffp-contract=off so that no multiply-add is fused behind our back.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mergenet_hip.h"

#define MN_EPS32 1.1920928955078125e-07f   /* 2^-23, np.finfo(np.float32).eps */
#define MN_EMPTY 0xFFFFFFFFFFFFFFFFull
#define MN_FIX_ONE 1073741824.0             /* 2^30: fixed-point scale of log-odds sums */

typedef unsigned long long u64;
typedef long long i64;

struct ImgParams {
  int H, W, N, C, O;
  float sdb, omf, bias;
  int variant, clip;
  int banded;          // 1: XCD-banded tile order in the neighbour-reading kernels (mn_xcd_tile)
  int djmin, djmax;    // smallest / largest column offset of the list
  float vmin_first;    // pixel-level edges with a raw sameness value below this cannot reach priority >= 0
  float sep_hi, sep_lo; // components mode: an edge inside a component needs a sameness value >= sep_hi,
                        // one between components <= sep_lo (0.5 widened by the float32 rounding margin)
  const float* cls;    // [C][N] class probabilities (borrowed)
  const float* same;   // [O][N] sameness probabilities (borrowed)
  int di[MN_MAX_OFFSETS];
  int dj[MN_MAX_OFFSETS];
};

// Per-object state.  Object id = pixel id of the surviving pixel (segment.cc:197-206).
// lpsum is filled lazily: an object that never merged reads its class log-probs straight
// from the class planes (lpvalid == 0), so phase A never writes a [C][N] table.
struct ObjState {
  unsigned char* ocls;     // [N] current class            (Object::object_class)
  int* osize;              // [N] pixel count              (Object::pixels.size())
  int* parent;             // [N] absorbed -> survivor; self for live objects
  float* lpsum;            // [C][N] summed class log-probs (Object::class_logprobs)
  unsigned char* lpvalid;  // [N]
};

__device__ __forceinline__ float mn_clip(float v) {
  return fminf(fmaxf(v, MN_EPS32), 1.0f - MN_EPS32);
}

// log(1 - v) in float32 without the cancellation of a plain 1.0f - v: d = fl(1 - v),
// e = (1 - d) - v is the exact rounding residual (Fast2Sum, 1 >= v), log(d + e) ~ log d + e/d.
// The reference computes (float)log(1.0 - (double)v) (segment.cc:34).
__device__ __forceinline__ float mn_log1m(float v) {
  const float d = 1.0f - v;
  const float e = (1.0f - d) - v;
  return logf(d) + e / d;
}

// same_different_bias applied on load instead of rewriting the plane (segment.cc:183-195).
__device__ __forceinline__ float mn_same_value(const ImgParams& P, float v) {
  if (P.clip) v = mn_clip(v);
  if (P.sdb != 0.0f) {
    const float logit = (logf(v) - mn_log1m(v)) + P.sdb;
    v = 1.0f / (1.0f + expf(-logit));
  }
  return v;
}

__device__ __forceinline__ float mn_ld_class(const ImgParams& P, int c, int p) {
  float v = P.cls[(size_t)c * P.N + p];
  if (P.clip) v = mn_clip(v);
  return v;
}

// log-odds of one edge (segment.cc:33-36), quantised to the fixed-point unit used for sums so
// that every kernel sees the same value for the same edge.
__device__ __forceinline__ i64 mn_edge_fixed(float v) {
  const float oml = logf(v) - mn_log1m(v);
  return __float2ll_rn(oml * (float)MN_FIX_ONE);   // float * 2^30 is exact: same value as in double
}
__device__ __forceinline__ float mn_fixed_to_float(i64 s) {
  return (float)((double)s * (1.0 / MN_FIX_ONE));
}

__device__ __forceinline__ float mn_obj_lp(const ImgParams& P, const ObjState& S, bool valid,
                                           int obj, int c) {
  return valid ? S.lpsum[(size_t)c * P.N + obj] : logf(mn_ld_class(P, c, obj));
}

// Priority of the record (u, v), u < v, with summed log-odds `oml`
// (ComputeClassDeltaLogprob + UpdateMergePriority, segment.cc:107-150; the Python variant
// utils/segmenter.py:179-193 differs in the denominator and in where the bias sits).
// *gain_pos reports whether the likelihood gain itself (numerator without the bias) is > 0.
__device__ __forceinline__ float mn_score(const ImgParams& P, const ObjState& S, int u, int v,
                                          float oml, int* merged_cls, bool* gain_pos) {
  const int cu = S.ocls[u], cv = S.ocls[v];
  float cdl = 0.0f;
  int mc = cu;
  if (cu != cv) {
    const bool vu = S.lpvalid[u] != 0, vv = S.lpvalid[v] != 0;
    float best = 0.0f, lu = 0.0f, lv = 0.0f;
    for (int c = 0; c < P.C; c++) {
      const float a = mn_obj_lp(P, S, vu, u, c);
      const float b = mn_obj_lp(P, S, vv, v, c);
      const float j = a + b;
      if (c == 0 || j > best) { best = j; mc = c; }
      if (c == cu) lu = a;
      if (c == cv) lv = b;
    }
    cdl = (best - lu) - lv;
  }
  *merged_cls = mc;
  const int nu = S.osize[u], nv = S.osize[v];
  const float num = oml * P.omf + cdl;
  *gain_pos = num > 0.0f;
  if (P.variant == MN_VARIANT_CSEGMENT) {
    const float den = (float)(nu + nv);
    return num / den + P.bias;
  }
  const float den = (float)nu * (float)nv;
  return (num + P.bias) / den;
}

// The same priority, computed by a whole wave for one record (u, v, oml uniform over the wave).
// Round trip 1: all six object fields at once.  Only if the classes differ, round trip 2: lane c
// takes class c and requests BOTH sources of its log-prob (summed table and class plane) for both
// objects before any is used; the first maximum is found with log2(C) shuffle steps.  The
// sequential finisher so pays one or two global round trips per score instead of two to four.
// The arithmetic per class, the first-maximum rule and (best - lu) - lv are those of mn_score:
// the results are bit-identical.  `fields`, if given, receives {n_u, n_v, valid_u, valid_v}.
__device__ __forceinline__ float mn_score_wave(const ImgParams& P, const ObjState& S, int u, int v,
                                               float oml, int* merged_cls, bool* gain_pos,
                                               int* fields = nullptr) {
  const int lane = threadIdx.x & 63;
  const int cu = S.ocls[u], cv = S.ocls[v];
  const bool vu = S.lpvalid[u] != 0, vv = S.lpvalid[v] != 0;
  const int nu = S.osize[u], nv = S.osize[v];
  if (fields) { fields[0] = nu; fields[1] = nv; fields[2] = vu ? 1 : 0; fields[3] = vv ? 1 : 0; }
  float cdl = 0.0f;
  int mc = cu;
  if (cu != cv) {                                     // uniform
    float best = 0.0f, lu = 0.0f, lv = 0.0f;
    bool have_best = false;
    int width = 64;                                   // lanes that hold a class in a chunk
    while (width > 1 && (width >> 1) >= P.C) width >>= 1;
    for (int c0 = 0; c0 < P.C; c0 += 64) {
      const int c = c0 + lane;
      float a = 0.0f, b = 0.0f;
      if (c < P.C) {
        const float as = S.lpsum[(size_t)c * P.N + u], ap = mn_ld_class(P, c, u);
        const float bs = S.lpsum[(size_t)c * P.N + v], bp = mn_ld_class(P, c, v);
        a = vu ? as : logf(ap);
        b = vv ? bs : logf(bp);
      }
      // first maximum over the classes of this chunk: highest j, lowest c among equals
      float j = a + b;
      int jc = c < P.C ? c : 0x7FFFFFFF;
      bool ok = c < P.C;
      for (int off = width >> 1; off > 0; off >>= 1) {
        const float oj = __shfl_xor(j, off);
        const int oc = __shfl_xor(jc, off);
        const bool ook = __shfl_xor((int)ok, off) != 0;
        const bool take = ook && (!ok || oj > j || (oj == j && oc < jc));
        if (take) { j = oj; jc = oc; ok = true; }
      }
      j = __shfl(j, 0); jc = __shfl(jc, 0);           // lanes beyond `width` did not take part
      if (!have_best || j > best) { best = j; mc = jc; have_best = true; }
      if (cu >= c0 && cu < c0 + 64) lu = __shfl(a, cu - c0);
      if (cv >= c0 && cv < c0 + 64) lv = __shfl(b, cv - c0);
    }
    cdl = (best - lu) - lv;
  }
  *merged_cls = mc;
  const float num = oml * P.omf + cdl;
  *gain_pos = num > 0.0f;
  if (P.variant == MN_VARIANT_CSEGMENT) {
    const float den = (float)(nu + nv);
    return num / den + P.bias;
  }
  const float den = (float)nu * (float)nv;
  return (num + P.bias) / den;
}

// (priority, partner) packed so that an unsigned max picks the highest priority and, among
// equal priorities, the LOWEST partner id.  Only priorities >= 0 are packed.  Object ids are
// below 2^28, which leaves bit 31 of the low word for a flag ("likelihood gain > 0") that both
// endpoints of a record compute identically.
__device__ __forceinline__ u64 mn_pack(float prio, int partner, bool gain_pos = false) {
  const unsigned bits = (prio == 0.0f) ? 0u : __float_as_uint(prio);
  return ((u64)bits << 32) | (gain_pos ? 0x80000000ull : 0ull) |
         (u64)(0x7FFFFFFFu - (unsigned)partner);
}
__device__ __forceinline__ int mn_pack_partner(u64 k) {
  return (int)(0x7FFFFFFFu - (unsigned)(k & 0x7FFFFFFFull));
}
__device__ __forceinline__ bool mn_pack_gain_pos(u64 k) { return (k & 0x80000000ull) != 0; }

__device__ __forceinline__ u64 mn_key(int a, int b) {
  const unsigned lo = (unsigned)min(a, b), hi = (unsigned)max(a, b);
  return ((u64)lo << 32) | (u64)hi;
}
__device__ __forceinline__ int mn_key_u(u64 k) { return (int)(k >> 32); }
__device__ __forceinline__ int mn_key_v(u64 k) { return (int)(k & 0xFFFFFFFFull); }

__device__ __forceinline__ unsigned mn_hash(u64 k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (unsigned)k;
}

// XCD-aware tile order for kernels whose neighbouring tiles re-read each other's rows (shifted
// neighbour loads).  Workgroups are dealt round-robin over the 8 XCDs, each with a private L2:
// with the identity mapping the rows a tile needs from its neighbours were fetched into ANOTHER
// XCD's L2.  Here XCD j walks the contiguous band of tiles [j*chunk, (j+1)*chunk) in order, so
// the shifted re-reads hit its own L2.  Launch 8*chunk blocks; returns -1 for padding blocks.
// Speed only: any placement gives the same result.
__device__ __forceinline__ int mn_xcd_tile(int ntiles, int banded) {
  if (!banded) return (int)blockIdx.x < ntiles ? (int)blockIdx.x : -1;
  const int chunk = (ntiles + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
  return ((int)(blockIdx.x >> 3) < chunk && tile < ntiles) ? tile : -1;
}
