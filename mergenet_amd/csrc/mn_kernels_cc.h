// mn_kernels_cc.h -- component contraction: phase 1 of the merge in one union-find sweep, for
// inputs where phase 1 is provably order-independent.
//
// Claim (csegment variant, object_merge_factor > 0, merge_logprob_bias >= 0).  Let the pixel
// graph be SIGN-SEPARABLE: the connected components of the edges with positive log-odds are such
// that (a) every edge inside a component has gain >= tau, (b) every edge between components has
// gain <= -tau, (c) all pixels of a component share one arg-max class; tau > 0 is the float32
// rounding margin of fill_params (2 N ulp(bias): the quotient below must not vanish in the sum).  Then at every moment of
// the reference's run (utils/csegment/segment.cc:539-727) every record between two sub-objects of
// one component scores  sum(log-odds)*omf / (n1+n2) + bias  >  bias  (class delta 0, segment.cc:
// 107-150) and every record between sub-objects of different components scores < bias (negative
// log-odds, class delta <= 0).  The queue pops in descending priority, so no cross record is
// popped while any intra record is alive: the reference first merges every component completely
// -- in whatever order -- and only then turns to the records between components.  The state at
// that moment (objects = components, one fully summed record per adjacent pair) is what this
// file builds directly.  The O sameness planes are read from HBM exactly ONCE:
//   mn_cc_sign     THE sweep over the class and sameness planes (4 B per value, the roofline-judged pass
//                  of this mode): per pixel a bit mask of its positive out-edges and one of its negative
//                  out-edges (4 + 4 B/pixel), the margin test of (a)/(b), the log sums of the certificate,
//                  per lane and class the fixed-point log of the product of its four pixels' class values;
//                  everything after this kernel works on the masks (mn_cc_cross reads the values of the
//                  negative edges again: 1.2 M of 21 M at 1024x2048);
//   mn_cc_tiles    16 x 64-pixel tiles labelled in LDS over the two unit offsets (mask bits); flat
//                  within the tile; accumulators of the tile roots cleared;
//   mn_cc_borders  unit-offset edges across tile borders; mn_cc_flatten  parent[p] = root, once;
//   mn_cc_hook     lock-free union-find over the other offsets (mask bits), root = lowest pixel id
//                  of the component;
//   mn_cc_class_sums  the class pass: arg-max class of every pixel, component sizes, class
//                  log-prob sums and class range (block table in LDS, 64-bit fixed-point atomics);
//                  also leaves parent[] flat;
//   mn_cc_cross    the mask of negative out-edges -> records between components (block scan, LDS queue,
//                  block table in LDS, then the global table); a negative edge inside a component fails (a);
//   mn_cc_finish   fixed-point sums -> float object state, condition (c), list of component roots.
// The second phase (records between components, where the bias lets a 1.6 M-pixel background
// swallow small instances) is then run by the sequential finisher in the reference's order,
// starting from freshly scored records.  If (a)-(c) fail the caller falls back to the general
// rounds: a spurious positive link across a boundary joins two components, which then contain
// negative edges, so the check -- not luck -- keeps this shortcut safe.
#pragma once

static_assert(MN_MAX_OFFSETS <= 32, "edge masks are 32-bit words: one bit per offset");

#include "mn_device.h"
#include "mn_kernels_merge.h"

#define MN_LP_FIX 16777216.0     /* 2^24: fixed-point scale of class log-prob sums.  |log p| <= 16, so a
                                    value fits an int32 (ONE v_cvt_i32_f32; a float -> int64 conversion
                                    is a dozen instructions and the sweep is VALU-bound), four of them
                                    still do, and 2^-24 is finer than a float log of magnitude >= 0.5 */

__device__ __forceinline__ int mn_cc_find(int* __restrict__ parent, int x) {
  int p = parent[x];
  while (p != x) {
    const int g = parent[p];
    if (g != p) parent[x] = g;      // path halving (benign race: only ever points further up)
    x = p;
    p = g;
  }
  return x;
}

struct __attribute__((packed, aligned(4))) mn_int4u { int x, y, z, w; };
__device__ __forceinline__ int4 mn_ld_int4_unaligned(const int* __restrict__ p) {
  const mn_int4u t = *reinterpret_cast<const mn_int4u*>(p);
  return make_int4(t.x, t.y, t.z, t.w);
}

// ---- the sweep over the class and sameness planes -------------------------------------------------
// One lane takes PX consecutive pixels (PX = 4 with 16-byte loads whenever N % 4 == 0 and W >= 4 -- with
// W % 4 != 0 one lane per row runs over the row's end: `straddle` --, else 1).  For every in-bounds
// (pixel, offset) pair the value decides: >= sep_hi the edge is positive (bit k of the pixel's positive
// mask), <= sep_lo it is negative (bit k of its negative mask), in between the map is not separable
// (margins: fill_params).  The negative edges -- a few per cent, along the instance boundaries -- are the
// only ones whose value is needed again (log-odds of the records between components): mn_cc_cross, on the
// side stream, enumerates the negative mask and reads those values again.  (Round 2 queued them in LDS and
// wrote a list from inside the sweep: two barriers at the end of every block and 168 MB of list regions
// per context; without it the sweep has neither LDS nor barrier: 37.7 instead of 43.8 us.)
// Algorithmic HBM bytes: 4 * (C + O) per pixel read (159.4 MB at 1024x2048, C = 9, O = 10).
//
// The pass is VALU-bound as soon as a value costs more than ~15 instructions (a wave64
// instruction takes 4 cycles; PMC: the first version ran 54 per value, 70 % VALU-busy), so:
//  * a wave whose pixels have every offset's column inside the image (all but the first and last
//    waves of a row) skips the per-value bounds tests: a row outside the image just loads 1.0;
//  * the certificate only needs  sum log v (inside) + sum log(1-v) (between) = sum log max(v, 1-v)
//    on a separable map, taken as the log of a PRODUCT per lane (factors in [0.5, 1], folded
//    every 80 factors): one max and one multiply per value instead of a log.
// PLAIN: no clip and no same_different_bias (the production setting), decided at compile time.
#define MN_CC_SIGN_THREADS 256
#define MN_CC_SIGN_G 5           /* offsets whose loads are in flight together */
#define MN_CC_EDGE_PIXBITS 26    /* components mode serves N <= 2^26 */

// Streaming accesses of the sweep: every plane value is read once and every output written once, so the
// loads / stores may carry the non-temporal hint (-DMN_NT_LOADS / -DMN_NT_STORES: measured)
typedef float mn_f4v __attribute__((ext_vector_type(4)));
typedef unsigned mn_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 mn_ld_stream4(const float* p) {
#ifdef MN_NT_LOADS
  const mn_f4v t = __builtin_nontemporal_load(reinterpret_cast<const mn_f4v*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void mn_st_stream(int* p, int v) {
#ifdef MN_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void mn_st_stream(uint4* p, uint4 v) {
#ifdef MN_NT_STORES
  mn_u4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<mn_u4v*>(p));
#else
  *p = v;
#endif
}

template <bool PLAIN>
__device__ __forceinline__ float mn_cc_value(const ImgParams& P, float v) {
  return PLAIN ? v : mn_same_value(P, v);
}

// CLS (PX == 4, N % 4 == 0): the lane also streams the C class planes of its four pixels first -- the
// affinity-scoring sweep then reads every input tensor, class and sameness, exactly once:
// 4 * (C + O) bytes per pixel (159.4 MB at 1024x2048, C = 9, O = 10).  Per pixel the arg-max class
// (first maximum of logf, settled on the values with an exact tie check: mn_cc_class_part), per
// lane and class ONE log of the product of the four values (2^-24 fixed point, gsum[c][lane]): what
// the class sums of the components need, 9 B/pixel instead of the 36 B/pixel planes (mn_cc_sums).
struct ClsOut {
  unsigned char* ocls; unsigned char* cls0; unsigned char* lpvalid; int* gsum; size_t gstride;
};

// first maximum of logf over the classes of one pixel (Object::Object, segment.cc:5-21)
__device__ __forceinline__ int mn_cc_argmax_logf(const ImgParams& P, int p) {
  float best = 0.0f;
  int b = 0;
  for (int c = 0; c < P.C; c++) {
    const float l = logf(mn_ld_class(P, c, p));
    if (c == 0 || l > best) { best = l; b = c; }
  }
  return b;
}

__device__ __forceinline__ void mn_cc_class_part(const ImgParams& P, const ClsOut& CO, int i) {
  // The arg-max is taken on the VALUES (logf is monotone); the reference's first-maximum rule on
  // logf values differs only if a class of LOWER index lies within rounding distance of the maximum
  // (logf may map both to one float): `prev` keeps the largest value below the current best's index
  // and such pixels redo their arg-max with logf.  Values are >= 2^-23 after the binding's clip, so
  // the product of four stays above 2^-92 (relative error 2e-7, far below the float32 accumulation
  // of the reference it stands for).
  float4 best, prev = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
  int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  float4 nxt = mn_ld_stream4(P.cls + 4 * (size_t)i);
  for (int c = 0; c < P.C; c++) {
    float4 v = nxt;
    if (c + 1 < P.C)
      nxt = mn_ld_stream4(P.cls + (size_t)(c + 1) * P.N + 4 * (size_t)i);
    if (P.clip) { v.x = mn_clip(v.x); v.y = mn_clip(v.y); v.z = mn_clip(v.z); v.w = mn_clip(v.w); }
    if (c == 0) {
      best = v;
    } else {
      if (v.x > best.x) { prev.x = best.x; best.x = v.x; b0 = c; }
      if (v.y > best.y) { prev.y = best.y; best.y = v.y; b1 = c; }
      if (v.z > best.z) { prev.z = best.z; best.z = v.z; b2 = c; }
      if (v.w > best.w) { prev.w = best.w; best.w = v.w; b3 = c; }
    }
    // float * 2^24 is exact: the term is the exact product rounded to the nearest integer
    mn_st_stream(&CO.gsum[(size_t)c * CO.gstride + i], __float2int_rn(logf((v.x * v.y) * (v.z * v.w)) * 16777216.0f));
  }
  // a lower class within 2^-18 of the maximum (logs of magnitude < 16 are 2^-20 apart at most, and
  // the GPU's logf is within an ulp of libm's): settle it the reference's way
  const float near = 1.0f - 3.814697265625e-06f;
  if (prev.x >= best.x * near) b0 = mn_cc_argmax_logf(P, 4 * i);
  if (prev.y >= best.y * near) b1 = mn_cc_argmax_logf(P, 4 * i + 1);
  if (prev.z >= best.z * near) b2 = mn_cc_argmax_logf(P, 4 * i + 2);
  if (prev.w >= best.w * near) b3 = mn_cc_argmax_logf(P, 4 * i + 3);
  uchar4 o;
  o.x = (unsigned char)b0; o.y = (unsigned char)b1; o.z = (unsigned char)b2; o.w = (unsigned char)b3;
  // (pure components mode: only the component roots' class and validity flag are ever read, and
  //  mn_cc_finish sets those: two byte planes fewer to write -- ocls / lpvalid null)
  if (CO.ocls) *reinterpret_cast<uchar4*>(CO.ocls + 4 * (size_t)i) = o;
  *reinterpret_cast<uchar4*>(CO.cls0 + 4 * (size_t)i) = o;
  if (CO.lpvalid) *reinterpret_cast<uchar4*>(CO.lpvalid + 4 * (size_t)i) = make_uchar4(0, 0, 0, 0);
}

template <int PX, bool PLAIN, bool CLS>
__global__ __launch_bounds__(MN_CC_SIGN_THREADS) void mn_cc_sign(
    ImgParams P, unsigned* __restrict__ bits, unsigned* __restrict__ negbits, int* __restrict__ violations,
    double* __restrict__ partial, ClsOut CO) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ngroups = (P.N + PX - 1) / PX;
  const int i = blockIdx.x * MN_CC_SIGN_THREADS + threadIdx.x;
  const bool live = i < ngroups;
  const int p0 = live ? PX * i : 0;
  const int r = p0 / P.W, c0 = p0 - r * P.W;
  if constexpr (CLS) { if (live) mn_cc_class_part(P, CO, i); }
  float f = 1.0f;
  double t_sum = 0.0;
  unsigned m[PX], ng[PX];                                // positive / negative out-edges per pixel
#pragma unroll
  for (int j = 0; j < PX; j++) { m[j] = 0u; ng[j] = 0u; }
  unsigned inmask[PX];                                   // in-bounds out-edges per pixel
#pragma unroll
  for (int j = 0; j < PX; j++) inmask[j] = 0u;
  constexpr int G = MN_CC_SIGN_G;
  // every pixel of every lane of this wave has all offsets' columns inside the image?
  const bool colsafe = live && c0 + P.djmin >= 0 && c0 + PX - 1 + P.djmax < P.W;
  const bool fast = __all(colsafe);
  // W % 4 != 0 (with N % 4 == 0): the four pixels of one lane per row run over the row's end into the
  // next row -- its planes are still read by aligned 16-byte loads, the bounds tests take each pixel's
  // own row (such a lane is never `colsafe`, so only the slow branch below sees it)
  const bool straddle = PX == 4 && live && c0 + PX - 1 >= P.W;
  unsigned rowmask = 0u;                                 // offsets whose row is inside the image
  for (int k0 = 0; k0 < P.O; k0 += G) {
    float v[G][PX];
    int first[G];                                       // column of the first neighbour, or INT_MIN
    int rin[G];                                         // bit 0: the lane's row + di is inside, bit 1: the next row + di
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int k = k0 + g;
      first[g] = INT_MIN;
      rin[g] = 0;
#pragma unroll
      for (int j = 0; j < PX; j++) v[g][j] = 1.0f;      // neutral: factor 1, its bits are masked off
      if (live && k < P.O) {
        const int rr = r + P.di[k];
        const bool in1 = rr >= 0 && rr < P.H, in2 = straddle && rr + 1 >= 0 && rr + 1 < P.H;
        if (in1 || in2) {
          first[g] = c0 + P.dj[k];
          rin[g] = (in1 ? 1 : 0) | (in2 ? 2 : 0);
          if (in1) rowmask |= 1u << k;
          if (PX == 4) {
            const float4 t = mn_ld_stream4(P.same + (size_t)k * P.N + p0);
            v[g][0] = t.x; v[g][1 % PX] = t.y; v[g][2 % PX] = t.z; v[g][3 % PX] = t.w;
          } else {
            v[g][0] = P.same[(size_t)k * P.N + p0];
          }
        }
      }
    }
    if (fast) {
#pragma unroll
      for (int g = 0; g < G; g++) {
        // (a group may run past the last offset: a phantom offset k >= 32 must not wrap onto bits 0..2,
        //  where its neutral value 1.0 would forge positive edges -- MN_MAX_OFFSETS is 32)
        const unsigned bit = (k0 + g < P.O) ? (1u << ((k0 + g) & 31)) : 0u;
#pragma unroll
        for (int j = 0; j < PX; j++) {
          const float x = (PLAIN || first[g] != INT_MIN) ? mn_cc_value<PLAIN>(P, v[g][j]) : 1.0f;
          m[j] |= (x >= P.sep_hi) ? bit : 0u;
          ng[j] |= (x <= P.sep_lo) ? bit : 0u;
          f *= fmaxf(x, 1.0f - x);
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < G; g++) {
        // (a group may run past the last offset: a phantom offset k >= 32 must not wrap onto bits 0..2,
        //  where its neutral value 1.0 would forge positive edges -- MN_MAX_OFFSETS is 32)
        const unsigned bit = (k0 + g < P.O) ? (1u << ((k0 + g) & 31)) : 0u;
#pragma unroll
        for (int j = 0; j < PX; j++) {
          // first[g] == INT_MIN (row outside / no such offset) fails the column test; a pixel past the
          // row's end belongs to the next row (straddling lane)
          const bool second = straddle && c0 + j >= P.W;
          const bool inb = ((rin[g] >> (second ? 1 : 0)) & 1) &&
                           (unsigned)(first[g] + j - (second ? P.W : 0)) < (unsigned)P.W;
          const float x = inb ? mn_cc_value<PLAIN>(P, v[g][j]) : 1.0f;
          inmask[j] |= inb ? bit : 0u;
          m[j] |= (x >= P.sep_hi) ? bit : 0u;
          ng[j] |= (x <= P.sep_lo) ? bit : 0u;
          f *= fmaxf(x, 1.0f - x);
        }
      }
    }
    if (((k0 / G) & 3) == 3 || k0 + G >= P.O) {         // at most 4 * G * PX = 80 factors per product
      t_sum += (double)logf(f);
      f = 1.0f;
    }
  }
  int bad = 0;
#pragma unroll
  for (int j = 0; j < PX; j++) {
    const unsigned in = fast ? rowmask : inmask[j];
    m[j] &= in;
    ng[j] &= in;
    bad += __popc(in & ~(m[j] | ng[j]));                // inside the rounding margin of 0.5
  }
  // Positive and negative out-edges per pixel, 4 B each.  The negative ones become records between
  // components in mn_cc_cross, which reads their values again (they cluster along the instance borders: a few
  // MB).  Until round 3 the sweep itself queued them in LDS behind a block-wide scan and wrote a list of
  // (edge, log-odds): two barriers and a dependent pass at the end of every block, 6.5 of 43.8 us (the sweep
  // without it: 37.3 us; a kernel that only moves the sweep's bytes: 34.3 us -- tools/stream_ceiling.hip).
  if (live) {
    if (PX == 4) {
      mn_st_stream(reinterpret_cast<uint4*>(bits + p0), make_uint4(m[0], m[1 % PX], m[2 % PX], m[3 % PX]));
      mn_st_stream(reinterpret_cast<uint4*>(negbits + p0), make_uint4(ng[0], ng[1 % PX], ng[2 % PX], ng[3 % PX]));
    } else {
      bits[p0] = m[0];
      negbits[p0] = ng[0];
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    bad += __shfl_xor(bad, off);
    t_sum += __shfl_xor(t_sum, off);
  }
  if (lane == 0) {
    if (bad) atomicAdd(violations, bad);
    const size_t waveg = (size_t)blockIdx.x * (MN_CC_SIGN_THREADS / 64) + wave;
    partial[waveg * 2] = t_sum;                         // (inside + between; the split is not needed)
    partial[waveg * 2 + 1] = 0.0;                       // wave order: the sum is reproducible
  }
}

// What the sweep leaves, for the parity test of the sweep itself (mn_sweep_device): the negative
// out-edges as a dense [O][N] array of log-odds (the value mn_cc_cross works with).
__global__ __launch_bounds__(256) void mn_cc_export_neg(ImgParams P, const unsigned* __restrict__ negbits,
                                                        float* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P.N) return;
  unsigned left = negbits[p];
  while (left) {
    const int k = __ffs((int)left) - 1;
    left &= left - 1u;
    const float x = mn_same_value(P, P.same[(size_t)k * P.N + p]);
    out[(size_t)k * P.N + p] = logf(x) - mn_log1m(x);
  }
}

// ---- labelling on the 4 B/pixel masks -----------------------------------------------------------
// Tile stage: a block owns a tile of 16 rows x 64 columns and labels it in LDS.
//  1. rows, no atomics: a wave covers the 64 pixels of one tile row; lanes joined by positive edges
//     of the horizontal unit offset (bit kh, direction +1 column) form runs, and every pixel
//     points at the first pixel of its run (one ballot and bit arithmetic);
//  2. columns: positive edges of the vertical unit offset (bit kv, direction dv = +-1 row) that
//     stay inside the tile are united by a union-find on the LDS labels (one lane per distinct
//     pair of roots in a wave);
//  3. the flattened labels go to `parent` as pixel ids.  Local order (row, column) is the global
//     pixel order, so "larger root under smaller" keeps holding across stages.
#ifndef MN_CC_TILE_ROWS
#define MN_CC_TILE_ROWS 16    /* rows of a labelling tile (x 64 columns = the block of mn_cc_tiles) */
#endif
//  4. every pixel's component size starts at 0 and the accumulators of the TILE roots are cleared
//     (class sums -- only the roots' slots of the C planes are ever used -- and class range): the
//     stages that follow only remove roots, so the final roots are among them.
__global__ __launch_bounds__(MN_CC_TILE_ROWS * 64) void mn_cc_tiles(ImgParams P, const unsigned* __restrict__ bits,
                                                    int* __restrict__ parent, int kh, int kv, int dv,
                                                    int* __restrict__ osize, i64* __restrict__ lp_acc,
                                                    int* __restrict__ clsmin, int* __restrict__ clsmax,
                                                    unsigned char* __restrict__ cand) {
  __shared__ int lab[MN_CC_TILE_ROWS * 64];
  const int t = threadIdx.x, lane = t & 63, i = t >> 6;
  const int r = (int)blockIdx.y * MN_CC_TILE_ROWS + i, c = (int)blockIdx.x * 64 + lane;
  const bool in = r < P.H && c < P.W;
  const int p = in ? r * P.W + c : 0;
  const unsigned b = in ? bits[p] : 0u;                // a set bit implies an in-bounds neighbour
  const bool link = kh >= 0 && ((b >> kh) & 1u) && lane < 63;   // next pixel in the same tile row
  const int ni = i + dv;
  const bool vlink = kv >= 0 && ((b >> kv) & 1u) && ni >= 0 && ni < MN_CC_TILE_ROWS;
  const u64 m = __ballot(link);
  // run start = one past the highest lane below `lane` that has NO link to its successor
  const u64 below = lane ? (~m & ((1ull << lane) - 1ull)) : 0ull;
  const int start = below ? (64 - __clzll((long long)below)) : 0;
  lab[t] = i * 64 + start;
  __syncthreads();
  int a = 0, bb = 0;
  bool want = false;
  if (vlink) {
    a = mn_cc_find(lab, t);
    bb = mn_cc_find(lab, ni * 64 + lane);
    want = a != bb;
  }
  u64 todo = __ballot(want);
  const u64 key = mn_key(a, bb);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const u64 k0v = ((u64)(unsigned)__shfl((int)(key >> 32), first) << 32) |
                    (u64)(unsigned)__shfl((int)(key & 0xFFFFFFFFull), first);
    const bool mine = want && key == k0v;
    if (lane == first) {
      while (a != bb) {                                 // hook the larger root under the smaller
        if (a < bb) { const int x = a; a = bb; bb = x; }
        const int old = atomicMin(&lab[a], bb);
        if (old == a) break;
        a = mn_cc_find(lab, old);
        bb = mn_cc_find(lab, bb);
      }
    }
    todo &= ~__ballot(mine);
  }
  __syncthreads();
  if (!in) return;
  int x = t;
  while (lab[x] != x) x = lab[x];
  parent[p] = ((int)blockIdx.y * MN_CC_TILE_ROWS + (x >> 6)) * P.W + (int)blockIdx.x * 64 + (x & 63);
  cand[p] = (x == t) ? 1 : 0;               // the only pixels that can be a component root
  if (x == t) {
    osize[p] = 0;                           // (sizes are only ever read at roots: 8 MB fewer to write)
    for (int c = 0; c < P.C; c++) lp_acc[(size_t)c * P.N + p] = 0;
    clsmin[p] = 255;
    clsmax[p] = 0;
  }
}

// one lane per distinct pair of roots of a wave does the union (the 64 pixels of a wave mostly ask
// for the same few unions: runs of a row against the runs of another row)
__device__ __forceinline__ void mn_cc_wave_union(int* __restrict__ parent, bool want, int a, int b) {
  const int lane = threadIdx.x & 63;
  u64 todo = __ballot(want);
  const u64 key = mn_key(a, b);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const u64 k0v = ((u64)(unsigned)__shfl((int)(key >> 32), first) << 32) |
                    (u64)(unsigned)__shfl((int)(key & 0xFFFFFFFFull), first);
    const bool mine = want && key == k0v;
    if (lane == first) {
      while (a != b) {                                  // hook the larger root under the smaller
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[a], b);
        if (old == a) break;
        a = mn_cc_find(parent, old);
        b = mn_cc_find(parent, b);
      }
    }
    todo &= ~__ballot(mine);
  }
}

// Border stage: after mn_cc_tiles the only unit-offset edges still open are those that cross a
// tile border, and the 16 (or 64) edges of one border segment almost always ask for the same
// union.  One block per tile: wave 0 takes the 16 edges across the tile's right border, wave 1 the
// 64 edges across its lower (dv = +1) or upper (dv = -1) border.
__global__ __launch_bounds__(128) void mn_cc_borders(ImgParams P, const unsigned* __restrict__ bits,
                                                     int* __restrict__ parent, int kh, int kv, int dv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (int)blockIdx.y * MN_CC_TILE_ROWS, c0 = (int)blockIdx.x * 64;
  int p = -1, q = -1;
  if (wave == 0) {                                   // right border: (r0 + lane, c0 + 63) -> next column
    const int r = r0 + lane, c = c0 + 63;
    if (kh >= 0 && lane < MN_CC_TILE_ROWS && r < P.H && c + 1 < P.W) {
      p = r * P.W + c;
      if ((bits[p] >> kh) & 1u) q = p + 1;
    }
  } else {                                           // the border the vertical offset crosses
    const int r = dv > 0 ? r0 + MN_CC_TILE_ROWS - 1 : r0, c = c0 + lane;
    if (kv >= 0 && r < P.H && r + dv >= 0 && r + dv < P.H && c < P.W) {
      p = r * P.W + c;
      if ((bits[p] >> kv) & 1u) q = p + dv * P.W;
    }
  }
  int a = 0, b = 0;
  bool want = false;
  if (q >= 0) {
    a = mn_cc_find(parent, p);
    b = mn_cc_find(parent, q);
    want = a != b;
  }
  mn_cc_wave_union(parent, want, a, b);
}

// parent[p] = root, once, between the border stage and the sweep over the other offsets (4 pixels
// per lane; every pixel already points at its tile root, so the chase is one or two steps)
__global__ __launch_bounds__(256) void mn_cc_flatten(int N, int* __restrict__ parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = N >> 2;
  if (i < n4) {
    int4 r = *reinterpret_cast<const int4*>(parent + 4 * (size_t)i);
    int4 o = r;
    int x = r.x; while (parent[x] != x) x = parent[x]; o.x = x;
    if (r.y == r.x) o.y = o.x; else { x = r.y; while (parent[x] != x) x = parent[x]; o.y = x; }
    if (r.z == r.x) o.z = o.x; else { x = r.z; while (parent[x] != x) x = parent[x]; o.z = x; }
    if (r.w == r.x) o.w = o.x; else { x = r.w; while (parent[x] != x) x = parent[x]; o.w = x; }
    if (o.x != r.x || o.y != r.y || o.z != r.z || o.w != r.w)
      *reinterpret_cast<int4*>(parent + 4 * (size_t)i) = o;
  }
  if (i < N - (n4 << 2)) {
    const int p = (n4 << 2) + i;
    int x = p; while (parent[x] != x) x = parent[x];
    parent[p] = x;
  }
}

// The offsets of `kmask` (all but the unit offsets the tile and border stages took): after the
// flatten nearly every positive edge finds equal parents at both ends and costs two L2 reads.
template <int PX>
__global__ __launch_bounds__(256) void mn_cc_hook(ImgParams P, const unsigned* __restrict__ bits,
                                                  int* __restrict__ parent, unsigned kmask) {
  const int ngroups = (P.N + PX - 1) / PX;
  const int tile = mn_xcd_tile((ngroups + 255) >> 8, P.banded);
  const int i = tile < 0 ? ngroups : tile * 256 + threadIdx.x;
  const bool live = i < ngroups;
  const int p0 = live ? PX * i : 0;
  const int r = p0 / P.W, c0 = p0 - r * P.W;
  unsigned b[PX];
  int own[PX];
#pragma unroll
  for (int j = 0; j < PX; j++) { b[j] = 0u; own[j] = 0; }
  if (live) {
    if (PX == 4) {
      const uint4 t = *reinterpret_cast<const uint4*>(bits + p0);
      b[0] = t.x; b[1 % PX] = t.y; b[2 % PX] = t.z; b[3 % PX] = t.w;
      const int4 o = *reinterpret_cast<const int4*>(parent + p0);
      own[0] = o.x; own[1 % PX] = o.y; own[2 % PX] = o.z; own[3 % PX] = o.w;
    } else {
      b[0] = bits[p0];
      own[0] = parent[p0];
    }
  }
  unsigned any = 0u;
#pragma unroll
  for (int j = 0; j < PX; j++) any |= b[j];
  any &= kmask;
  constexpr int G = 4;                        // offsets whose loads are in flight together
  for (int k0 = 0; k0 < P.O; k0 += G) {
    if (__ballot((any >> k0) & ((1u << G) - 1u)) == 0) continue;          // uniform
    int rq[G][PX];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int k = k0 + g;
#pragma unroll
      for (int j = 0; j < PX; j++) rq[g][j] = own[j];
      if (k < P.O && ((any >> k) & 1u)) {
        const long long q0 = (long long)(r + P.di[k]) * P.W + c0 + P.dj[k];
        if (PX == 4 && q0 >= 0 && q0 + 3 < P.N) {
          const int4 t = mn_ld_int4_unaligned(parent + q0);
          rq[g][0] = t.x; rq[g][1 % PX] = t.y; rq[g][2 % PX] = t.z; rq[g][3 % PX] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < PX; j++)
            if (((b[j] >> k) & 1u) && q0 + j >= 0 && q0 + j < P.N) rq[g][j] = parent[q0 + j];
        }
      }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int k = k0 + g;
      if (k >= P.O) break;                    // uniform
      if (__ballot((any >> k) & 1u) == 0) continue;
#pragma unroll
      for (int j = 0; j < PX; j++) {
        bool want = false;
        int a = 0, bb = 0;
        if (((b[j] & kmask) >> k) & 1u) {     // a set bit implies an in-bounds neighbour
          if (rq[g][j] != own[j]) {           // equal parents: already one set, nothing to do
            const int q = p0 + j + P.di[k] * P.W + P.dj[k];
            a = mn_cc_find(parent, p0 + j);
            bb = mn_cc_find(parent, q);
            want = a != bb;
          }
        }
        mn_cc_wave_union(parent, want, a, bb);
      }
    }
  }
}

// ---- cores: the contraction as the FIRST step of the general rounds --------------------------------
// A map that is not sign-separable as a whole still is inside its instances: a pixel is CLEAN when
// every in-bounds edge of a SHORT offset it takes part in (as source or as target) is positive beyond
// the margin and joins two pixels of one arg-max class.  The clean pixels joined by positive same-class
// edges (all offsets) between clean pixels form the cores; a core inside which some edge is not
// positive is condemned and falls apart again (mn_core_check).  (With all offsets counted as short --
// mn_options::core_radius < 0 -- a clean pixel has no non-positive edge at all and no core can be
// condemned; that leaves a fringe as wide as the LONGEST offset around every instance, 40 pixels at the
// benchmark's offsets, where a short radius leaves the few pixels the maps are unsure about -- and
// moves further from the reference's order)  Inside a core every record between sub-objects is positive with class
// delta 0 whatever the order (the argument of conditions (a) and (c) above), so the reference merges
// each core completely sooner or later; the rounds start from the cores plus the single pixels of the
// fringe (the band of the largest offset's width along instance boundaries, and whatever noise
// touched) instead of from single pixels everywhere: a quarter of the records at 1024x2048.  What it
// changes is WHEN a fringe pixel meets the pixels of a core (as one object instead of as growing
// sub-objects): exact on sign-separable maps (positive records all precede the others), one more
// approximation of the reference's order elsewhere, like the rounds themselves.
__global__ __launch_bounds__(256) void mn_core_clean(ImgParams P, const unsigned* __restrict__ bits,
                                                     const unsigned char* __restrict__ cls0,
                                                     unsigned char* __restrict__ clean, unsigned kshort) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  const int r = p / P.W, c = p - r * P.W;
  const unsigned b = bits[p];
  const unsigned char mine = cls0[p];
  bool ok = true;
  for (int k = 0; k < P.O; k++) {
    if (!((kshort >> k) & 1u)) continue;
    const int di = P.di[k], dj = P.dj[k];
    if (r + di >= 0 && r + di < P.H && c + dj >= 0 && c + dj < P.W) {
      const int q = p + di * P.W + dj;
      ok = ok && ((b >> k) & 1u) && cls0[q] == mine;
    }
    if (r - di >= 0 && r - di < P.H && c - dj >= 0 && c - dj < P.W) {
      const int q = p - di * P.W - dj;
      ok = ok && ((bits[q] >> k) & 1u) && cls0[q] == mine;
    }
  }
  clean[p] = ok ? 1 : 0;
}

// positive same-class out-edges (ALL offsets) between two clean pixels, in the format of the sign
// sweep's masks (a set bit implies an in-bounds neighbour): what the labelling stages run on
__global__ __launch_bounds__(256) void mn_core_bits(ImgParams P, const unsigned char* __restrict__ clean,
                                                    const unsigned* __restrict__ bits,
                                                    const unsigned char* __restrict__ cls0,
                                                    unsigned* __restrict__ bits2) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  unsigned m = 0u;
  if (clean[p]) {
    const unsigned b = bits[p];
    const unsigned char mine = cls0[p];
    const int r = p / P.W, c = p - r * P.W;
    for (int k = 0; k < P.O; k++) {
      if (!((b >> k) & 1u)) continue;                   // (a set bit implies an in-bounds neighbour)
      const int q = (r + P.di[k]) * P.W + c + P.dj[k];
      if (clean[q] && cls0[q] == mine) m |= 1u << k;
    }
  }
  bits2[p] = m;
}

// With `kshort` a subset of the offsets (the short ones), a core may hold two pixels joined by an edge
// that is NOT positive (a long offset reaching across a thin bridge): then the records inside the core
// are not all positive and the argument above fails -- for THAT core.  Every in-bounds edge between
// two pixels of one core must be in the core's mask; a core with one that is not is condemned.
__global__ __launch_bounds__(256) void mn_core_check(ImgParams P, const int* __restrict__ parent,
                                                     const unsigned* __restrict__ bits2,
                                                     unsigned char* __restrict__ condemned) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  const int root = parent[p];
  const unsigned b = bits2[p];
  const int r = p / P.W, c = p - r * P.W;
  for (int k = 0; k < P.O; k++) {
    if ((b >> k) & 1u) continue;
    const int rr = r + P.di[k], cc = c + P.dj[k];
    if (rr < 0 || rr >= P.H || cc < 0 || cc >= P.W) continue;
    if (parent[rr * P.W + cc] == root) { condemned[root] = 1; return; }
  }
}

// condemned cores fall apart into single pixels again (sizes 1, class sums read from the planes)
__global__ __launch_bounds__(256) void mn_core_dissolve(ImgParams P, ObjState S,
                                                        const unsigned char* __restrict__ condemned,
                                                        int* __restrict__ ndissolved) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  const int root = S.parent[p];
  if (!condemned[root]) return;
  // (the root's own entry is rewritten by the root's thread: every pixel of the core reads
  //  condemned[root], which nobody clears, and parent[p] only of itself)
  S.parent[p] = p;
  S.osize[p] = 1;
  S.lpvalid[p] = 0;
  if (p == root) *ndissolved = 1;
}

// in-bounds pixel edges between different objects: the number of insertions mn_build_from_pixels will
// make, which sizes its table (64 words, one atomic per block)
__global__ __launch_bounds__(256) void mn_count_cross_edges(ImgParams P, const int* __restrict__ parent,
                                                            unsigned* __restrict__ outside) {
  __shared__ int sh_n;
  if (threadIdx.x == 0) sh_n = 0;
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  int out = 0;
  if (p < P.N) {
    const int u = parent[p];
    const int r = p / P.W, c = p - r * P.W;
    for (int k = 0; k < P.O; k++) {
      const int rr = r + P.di[k], cc = c + P.dj[k];
      if (rr < 0 || rr >= P.H || cc < 0 || cc >= P.W) continue;
      out += parent[rr * P.W + cc] != u ? 1 : 0;
    }
  }
  for (int off = 32; off > 0; off >>= 1) out += __shfl_xor(out, off);
  if ((threadIdx.x & 63) == 0 && out) atomicAdd(&sh_n, out);
  __syncthreads();
  if (threadIdx.x == 0 && sh_n) atomicAdd(&outside[blockIdx.x & 63], (unsigned)sh_n);
}

// Class pass of components mode: one sweep over the C class planes gives every pixel its arg-max
// class (same rule as mn_class_pass: first maximum of logf) AND adds its class log-probs and its
// count to the sums of its component, and the component's class range (condition (c): one class
// per component <=> min == max).  The sweep also leaves parent[] flat (every pixel -> its root:
// no union happens after the hook stage) and clears lpvalid.  4 consecutive pixels per lane (16-byte loads); a lane whose
// four pixels share a root -- nearly all do -- issues one LDS atomic per class into the block's
// table (root -> C+1 fixed-point sums); the block then issues ONE global atomic per root and class:
// a 1.6 M-pixel background is a hot word for every wave of the image, and one word takes only ~88
// atomics/us.  64-bit fixed-point sums (2^-32) are order-independent.
#ifndef MN_CC_SUM_THREADS
#define MN_CC_SUM_THREADS 1024
#endif
#ifndef MN_CC_SUMS_G
#define MN_CC_SUMS_G 9         /* class sums of a lane requested together (mn_cc_sums) */
#endif
#ifndef MN_CC_SUMS_ITERS
#define MN_CC_SUMS_ITERS 1     /* chunks of 4096 pixels per block of mn_cc_sums */
#endif
#ifndef MN_CC_SUM_AHEAD
#define MN_CC_SUM_AHEAD 1   /* planes in flight ahead of the one in use: 1 at 1024 threads measured best (37.3 us by events; 2: 41.4, 3: 42.4; 512 threads: 43.7 / 40.8; 256: 47.6) -- occupancy matters more */
#endif
#ifndef MN_CC_SUMS_WAVE
#define MN_CC_SUMS_WAVE 1     /* a wave inside one component: ONE LDS atomic per class (0: one per 16-lane row) */
#endif
#define MN_CC_SUM_SLOTS 64
__device__ __forceinline__ int mn_lds_root_slot(int* s_root, int root) {
  unsigned h = ((unsigned)root * 2654435761u) >> 26;             // 6 bits
#pragma unroll 1
  for (int t = 0; t < MN_CC_SUM_SLOTS; t++) {
    int cur = s_root[h];                       // plain read first: the slot is usually there already
    if (cur == root) return (int)h;
    if (cur == -1) cur = atomicCAS(&s_root[h], -1, root);
    if (cur == -1 || cur == root) return (int)h;
    h = (h + 1) & (MN_CC_SUM_SLOTS - 1);
  }
  return -1;
}

__device__ __forceinline__ int mn_cc_root_ro(const int* __restrict__ parent, int x) {
  int p = parent[x];
  while (p != x) { x = p; p = parent[x]; }
  return x;
}

__device__ __forceinline__ void mn_cc_add(const ImgParams& P, const ObjState& S, int* s_root,
                                          u64* s_val, i64* __restrict__ lp_acc, int root, int c,
                                          int slot, i64 x) {
  if (slot >= 0) atomicAdd(&s_val[slot * (P.C + 1) + c], (u64)x);
  else if (c == P.C) atomicAdd(&S.osize[root], (int)x);
  else atomicAdd(reinterpret_cast<u64*>(&lp_acc[(size_t)c * P.N + root]), (u64)x);
}

__device__ __forceinline__ void mn_cc_cls(int* s_min, int* s_max, int* __restrict__ clsmin,
                                          int* __restrict__ clsmax, int root, int slot, int lo, int hi) {
  if (slot >= 0) {
    if (lo < s_min[slot]) atomicMin(&s_min[slot], lo);
    if (hi > s_max[slot]) atomicMax(&s_max[slot], hi);
  } else {
    atomicMin(&clsmin[root], lo);
    atomicMax(&clsmax[root], hi);
  }
}

__global__ __launch_bounds__(MN_CC_SUM_THREADS) void mn_cc_class_sums(
    ImgParams P, ObjState S, unsigned char* __restrict__ cls0, i64* __restrict__ lp_acc,
    int* __restrict__ clsmin, int* __restrict__ clsmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  u64* s_val = reinterpret_cast<u64*>(cc_smem);                   // [SLOTS][C+1], index C = count
  __shared__ int s_root[MN_CC_SUM_SLOTS];
  __shared__ int s_min[MN_CC_SUM_SLOTS];
  __shared__ int s_max[MN_CC_SUM_SLOTS];
  const int nval = MN_CC_SUM_SLOTS * (P.C + 1);
  for (int i = threadIdx.x; i < nval; i += MN_CC_SUM_THREADS) s_val[i] = 0;
  if (threadIdx.x < MN_CC_SUM_SLOTS) { s_root[threadIdx.x] = -1; s_min[threadIdx.x] = 255; s_max[threadIdx.x] = 0; }
  __syncthreads();
  const int n4 = P.N >> 2;
  const int i = blockIdx.x * MN_CC_SUM_THREADS + threadIdx.x;
  if (i < n4) {
    int4 r = *reinterpret_cast<const int4*>(S.parent + 4 * (size_t)i);
    // after the hook stage a parent may still be one or two steps from its root
    r.x = mn_cc_root_ro(S.parent, r.x);
    r.y = r.y == r.x ? r.x : mn_cc_root_ro(S.parent, r.y);
    r.z = r.z == r.x ? r.x : mn_cc_root_ro(S.parent, r.z);
    r.w = r.w == r.x ? r.x : mn_cc_root_ro(S.parent, r.w);
    *reinterpret_cast<int4*>(S.parent + 4 * (size_t)i) = r;
    *reinterpret_cast<uchar4*>(S.lpvalid + 4 * (size_t)i) = make_uchar4(0, 0, 0, 0);
    const bool same = r.x == r.y && r.x == r.z && r.x == r.w;
    const int s0 = mn_lds_root_slot(s_root, r.x);
    const int s1 = same ? s0 : mn_lds_root_slot(s_root, r.y);
    const int s2 = same ? s0 : mn_lds_root_slot(s_root, r.z);
    const int s3 = same ? s0 : mn_lds_root_slot(s_root, r.w);
    // The sweep is VALU-bound if every value takes a libm log (36 per lane, ~22 instructions each):
    //  * the arg-max is taken on the VALUES (logf is monotone); the reference's first-maximum rule
    //    on logf values differs only if a class of LOWER index lies within rounding distance of the
    //    maximum (logf may map both to one float) -- `prev` keeps the largest value below the
    //    current best's index and such pixels redo their arg-max with logf (mn_cc_argmax_logf);
    //  * a lane whose four pixels share a component -- nearly all -- adds  log(v0 v1 v2 v3)  per
    //    class: ONE log instead of four (values >= 2^-23 after the binding's clip, so the product
    //    stays above 2^-92; relative error 2e-7, far below the float32 accumulation it replaces).
    float4 best, prev = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    // the next plane's load is issued before this plane's values are used (one extra float4 of
    // registers: the kernel stays at two 1024-thread blocks per CU, which matters more here than
    // deeper staging -- three planes in flight at 85 VGPRs measured slower, and so did ALL nine
    // in flight at 82 VGPRs with the cheaper arithmetic of round 2: 37.9 against 30.4 us)
    // MN_CC_SUM_AHEAD planes of the lane are in flight: a wave's chain of dependent HBM round trips
    // is what the kernel's duration is made of (the grid is one residency of the chip)
    float4 ring[MN_CC_SUM_AHEAD];
#pragma unroll
    for (int a = 0; a < MN_CC_SUM_AHEAD; a++)
      ring[a] = a < P.C ? *reinterpret_cast<const float4*>(P.cls + (size_t)a * P.N + 4 * (size_t)i)
                        : make_float4(0.5f, 0.5f, 0.5f, 0.5f);
    for (int c0 = 0; c0 < P.C; c0 += MN_CC_SUM_AHEAD) {
#pragma unroll
    for (int a = 0; a < MN_CC_SUM_AHEAD; a++) {
      const int c = c0 + a;
      if (c >= P.C) break;
      float4 v = ring[a];
      if (c + MN_CC_SUM_AHEAD < P.C)
        ring[a] = *reinterpret_cast<const float4*>(P.cls + (size_t)(c + MN_CC_SUM_AHEAD) * P.N + 4 * (size_t)i);
      if (P.clip) { v.x = mn_clip(v.x); v.y = mn_clip(v.y); v.z = mn_clip(v.z); v.w = mn_clip(v.w); }
      if (c == 0) {
        best = v;
      } else {
        if (v.x > best.x) { prev.x = best.x; best.x = v.x; b0 = c; }
        if (v.y > best.y) { prev.y = best.y; best.y = v.y; b1 = c; }
        if (v.z > best.z) { prev.z = best.z; best.z = v.z; b2 = c; }
        if (v.w > best.w) { prev.w = best.w; best.w = v.w; b3 = c; }
      }
      // float * 2^24 is exact, so each term is the exact product rounded to the nearest integer
      if (same) {
        const float l = logf((v.x * v.y) * (v.z * v.w));
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, c, s0, (i64)__float2int_rn(l * (float)MN_LP_FIX));
      } else {
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, c, s0, (i64)__float2int_rn(logf(v.x) * (float)MN_LP_FIX));
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.y, c, s1, (i64)__float2int_rn(logf(v.y) * (float)MN_LP_FIX));
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.z, c, s2, (i64)__float2int_rn(logf(v.z) * (float)MN_LP_FIX));
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.w, c, s3, (i64)__float2int_rn(logf(v.w) * (float)MN_LP_FIX));
      }
    }
    }
    // a lower class within 2^-18 of the maximum (logs of magnitude < 16 are 2^-20 apart at most, and
    // the GPU's logf is within an ulp of libm's): settle it the reference's way
    const float near = 1.0f - 3.814697265625e-06f;
    if (prev.x >= best.x * near) b0 = mn_cc_argmax_logf(P, 4 * i);
    if (prev.y >= best.y * near) b1 = mn_cc_argmax_logf(P, 4 * i + 1);
    if (prev.z >= best.z * near) b2 = mn_cc_argmax_logf(P, 4 * i + 2);
    if (prev.w >= best.w * near) b3 = mn_cc_argmax_logf(P, 4 * i + 3);
    if (same) {
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, P.C, s0, 4);
    } else {
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, P.C, s0, 1);
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.y, P.C, s1, 1);
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.z, P.C, s2, 1);
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.w, P.C, s3, 1);
    }
    uchar4 o;
    o.x = (unsigned char)b0; o.y = (unsigned char)b1; o.z = (unsigned char)b2; o.w = (unsigned char)b3;
    *reinterpret_cast<uchar4*>(S.ocls + 4 * (size_t)i) = o;
    *reinterpret_cast<uchar4*>(cls0 + 4 * (size_t)i) = o;
    if (same) {
      mn_cc_cls(s_min, s_max, clsmin, clsmax, r.x, s0, min(min(b0, b1), min(b2, b3)), max(max(b0, b1), max(b2, b3)));
    } else {
      mn_cc_cls(s_min, s_max, clsmin, clsmax, r.x, s0, b0, b0);
      mn_cc_cls(s_min, s_max, clsmin, clsmax, r.y, s1, b1, b1);
      mn_cc_cls(s_min, s_max, clsmin, clsmax, r.z, s2, b2, b2);
      mn_cc_cls(s_min, s_max, clsmin, clsmax, r.w, s3, b3, b3);
    }
  }
  // tail pixels when N is not a multiple of 4: straight to the global sums
  const int tail0 = n4 << 2;
  if (i < P.N - tail0) {
    const int p = tail0 + i;
    const int root = mn_cc_root_ro(S.parent, p);
    S.parent[p] = root;
    S.lpvalid[p] = 0;
    float best = 0.0f;
    int b = 0;
    for (int c = 0; c < P.C; c++) {
      const float l = logf(mn_ld_class(P, c, p));
      if (c == 0 || l > best) { best = l; b = c; }
      mn_cc_add(P, S, s_root, s_val, lp_acc, root, c, -1, (i64)__float2int_rn(l * (float)MN_LP_FIX));
    }
    mn_cc_add(P, S, s_root, s_val, lp_acc, root, P.C, -1, 1);
    S.ocls[p] = (unsigned char)b;
    cls0[p] = (unsigned char)b;
    mn_cc_cls(s_min, s_max, clsmin, clsmax, root, -1, b, b);
  }
  __syncthreads();
  if (threadIdx.x < MN_CC_SUM_SLOTS && s_root[threadIdx.x] >= 0) {
    atomicMin(&clsmin[s_root[threadIdx.x]], s_min[threadIdx.x]);
    atomicMax(&clsmax[s_root[threadIdx.x]], s_max[threadIdx.x]);
  }
  for (int j = threadIdx.x; j < nval; j += MN_CC_SUM_THREADS) {
    const u64 v = s_val[j];
    if (v == 0) continue;
    const int slot = j / (P.C + 1), c = j - slot * (P.C + 1);
    mn_cc_add(P, S, s_root, s_val, lp_acc, s_root[slot], c, -1, (i64)v);
  }
}



// Sum of a 64-bit value over the 16 lanes of a DPP row (all 16 lanes get it): quad_perm xor 1, xor 2,
// then row_half_mirror and row_mirror -- data-parallel-primitive moves on the VALU, no LDS crossbar
// (a ds_bpermute-based __shfl_xor costs about what the atomics it is meant to save cost).
__device__ __forceinline__ i64 mn_row16_sum(i64 x) {
  int lo = (int)(x & 0xFFFFFFFFll), hi = (int)(x >> 32);
#define MN_DPP_STEP(CTRL)                                                                   \
  {                                                                                         \
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);              \
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);              \
    const i64 y = ((i64)hi << 32 | (i64)(unsigned)lo) + ((i64)hi2 << 32 | (i64)(unsigned)lo2); \
    lo = (int)(y & 0xFFFFFFFFll); hi = (int)(y >> 32);                                      \
  }
  MN_DPP_STEP(0xB1)     // quad_perm [1,0,3,2]
  MN_DPP_STEP(0x4E)     // quad_perm [2,3,0,1]
  MN_DPP_STEP(0x141)    // row_half_mirror
  MN_DPP_STEP(0x140)    // row_mirror
#undef MN_DPP_STEP
  return (i64)hi << 32 | (i64)(unsigned)lo;
}

// ... and over the four rows of a wave, from a value every lane of a row already holds: four lane reads on
// the scalar unit (no LDS, no further DPP)
__device__ __forceinline__ i64 mn_rows4_sum(i64 rs) {
  const int lo = (int)(rs & 0xFFFFFFFFll), hi = (int)(rs >> 32);
  i64 t = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    t += (i64)__builtin_amdgcn_readlane(hi, 16 * k) << 32 | (i64)(unsigned)__builtin_amdgcn_readlane(lo, 16 * k);
  return t;
}

__device__ __forceinline__ int mn_row16_min(int x) {
  x = min(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
  x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
  x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));
  x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));
  return x;
}

// one class plane of a lane whose four pixels lie in different components: per-pixel logs
__device__ __forceinline__ void mn_cc_sums_pixelwise(const ImgParams& P, const ObjState& S, int* s_root,
                                                  u64* s_val, i64* __restrict__ lp_acc, int4 r, int c,
                                                  int s0, int s1, int s2, int s3, float4 v) {
  const float vx = P.clip ? mn_clip(v.x) : v.x, vy = P.clip ? mn_clip(v.y) : v.y;
  const float vz = P.clip ? mn_clip(v.z) : v.z, vw = P.clip ? mn_clip(v.w) : v.w;
  mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, c, s0, (i64)__float2int_rn(logf(vx) * (float)MN_LP_FIX));
  mn_cc_add(P, S, s_root, s_val, lp_acc, r.y, c, s1, (i64)__float2int_rn(logf(vy) * (float)MN_LP_FIX));
  mn_cc_add(P, S, s_root, s_val, lp_acc, r.z, c, s2, (i64)__float2int_rn(logf(vz) * (float)MN_LP_FIX));
  mn_cc_add(P, S, s_root, s_val, lp_acc, r.w, c, s3, (i64)__float2int_rn(logf(vw) * (float)MN_LP_FIX));
}

// Class sums of the components from what the sweep left (CLS form of mn_cc_sign): per lane and
// class the fixed-point log of the product of its four values, per pixel the arg-max class.  A lane
// whose four pixels share a component -- nearly all -- adds its C numbers and its class range to
// the block's LDS table; a lane across a component boundary goes back to the class planes for
// per-pixel logs.  Leaves parent[] flat.  9 B/pixel read instead of the 36 B/pixel planes.
__global__ __launch_bounds__(MN_CC_SUM_THREADS) void mn_cc_sums(
    ImgParams P, ObjState S, const unsigned char* __restrict__ cls0, const int* __restrict__ gsum,
    size_t gstride, i64* __restrict__ lp_acc, int* __restrict__ clsmin, int* __restrict__ clsmax,
    const unsigned char* __restrict__ clean) {
  // `clean` (cores ahead of the rounds, else null): a pixel that is not clean is a component of its
  // own by construction -- size 1, class sums read from the planes when needed -- and takes no sums
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  u64* s_val = reinterpret_cast<u64*>(cc_smem);                   // [SLOTS][C+1], index C = count
  __shared__ int s_root[MN_CC_SUM_SLOTS];
  __shared__ int s_min[MN_CC_SUM_SLOTS];
  __shared__ int s_max[MN_CC_SUM_SLOTS];
  __shared__ int s_q[MN_CC_SUM_THREADS * MN_CC_SUMS_ITERS];      // (a lane queues at most once per chunk)
  __shared__ int s_qn;
  const int nval = MN_CC_SUM_SLOTS * (P.C + 1);
  for (int i = threadIdx.x; i < nval; i += MN_CC_SUM_THREADS) s_val[i] = 0;
  if (threadIdx.x == 0) s_qn = 0;
  if (threadIdx.x < MN_CC_SUM_SLOTS) { s_root[threadIdx.x] = -1; s_min[threadIdx.x] = 255; s_max[threadIdx.x] = 0; }
  __syncthreads();
  const int n4 = P.N >> 2;
  // grid-stride: MN_CC_SUMS_ITERS chunks per block, so that the block's table is flushed -- global
  // atomics on the few hot words of the large components -- once per several thousand pixels more
  for (int base = blockIdx.x * MN_CC_SUM_THREADS; base < n4; base += gridDim.x * MN_CC_SUM_THREADS) {
  const int i = base + (int)threadIdx.x;
  if (i < n4) {
    // every load of the lane that does not depend on another is requested first -- parents, arg-max
    // classes and the first MN_CC_SUMS_G class sums: the kernel is a chain of dependent round trips
    // (parent -> root -> ...), not a stream, and these used to queue up behind the chase
    constexpr int G = MN_CC_SUMS_G;
    int4 r = *reinterpret_cast<const int4*>(S.parent + 4 * (size_t)i);
    const int4 r_in = r;
    const uchar4 b = *reinterpret_cast<const uchar4*>(cls0 + 4 * (size_t)i);
    int g[G];
#pragma unroll
    for (int a = 0; a < G; a++) g[a] = a < P.C ? gsum[(size_t)a * gstride + i] : 0;
    // after the hook stage a parent may still be one or two steps from its root
    r.x = mn_cc_root_ro(S.parent, r.x);
    r.y = r.y == r.x ? r.x : mn_cc_root_ro(S.parent, r.y);
    r.z = r.z == r.x ? r.x : mn_cc_root_ro(S.parent, r.z);
    r.w = r.w == r.x ? r.x : mn_cc_root_ro(S.parent, r.w);
    // (after the flatten stage only the pixels the hook stage re-rooted are not flat: 8 MB fewer written)
    if (r.x != r_in.x || r.y != r_in.y || r.z != r_in.z || r.w != r_in.w)
      *reinterpret_cast<int4*>(S.parent + 4 * (size_t)i) = r;
    const bool same = r.x == r.y && r.x == r.z && r.x == r.w;
    const int s0 = mn_lds_root_slot(s_root, r.x);
    // a wave inside one component -- most are -- would put 64 atomics on ONE LDS address per class,
    // and same-address LDS atomics run one after the other: its 16-lane rows are summed on the VALU
    // first, 4 atomics per class
    const bool uni = __all(same && r.x == __shfl(r.x, 0) && i - (int)(threadIdx.x & 63) + 63 < n4);
    if (same) {
      const int lane = threadIdx.x & 63;
      for (int c0 = 0; c0 < P.C; c0 += G) {
        if (c0 > 0) {
#pragma unroll
          for (int a = 0; a < G; a++) g[a] = c0 + a < P.C ? gsum[(size_t)(c0 + a) * gstride + i] : 0;
        }
#pragma unroll
        for (int a = 0; a < G; a++) {
          if (c0 + a >= P.C) break;
          if (uni) {
#if MN_CC_SUMS_WAVE
            const i64 ws = mn_rows4_sum(mn_row16_sum((i64)g[a]));
            if (lane == 0) mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, c0 + a, s0, ws);
#else
            const i64 rs = mn_row16_sum((i64)g[a]);
            if ((lane & 15) == 0) mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, c0 + a, s0, rs);
#endif
          } else {
            mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, c0 + a, s0, (i64)g[a]);
          }
        }
      }
      const int lo1 = min(min((int)b.x, (int)b.y), min((int)b.z, (int)b.w));
      const int hi1 = max(max((int)b.x, (int)b.y), max((int)b.z, (int)b.w));
      if (uni) {
#if MN_CC_SUMS_WAVE
        int lo = mn_row16_min(lo1), hi = -mn_row16_min(-hi1);
        lo = min(min(__builtin_amdgcn_readlane(lo, 0), __builtin_amdgcn_readlane(lo, 16)),
                 min(__builtin_amdgcn_readlane(lo, 32), __builtin_amdgcn_readlane(lo, 48)));
        hi = max(max(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(hi, 16)),
                 max(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(hi, 48)));
        if (lane == 0) {
          mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, P.C, s0, 256);
          mn_cc_cls(s_min, s_max, clsmin, clsmax, r.x, s0, lo, hi);
        }
#else
        if ((lane & 15) == 0) mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, P.C, s0, 64);
        const int lo = mn_row16_min(lo1), hi = -mn_row16_min(-hi1);
        if ((lane & 15) == 0) mn_cc_cls(s_min, s_max, clsmin, clsmax, r.x, s0, lo, hi);
#endif
      } else {
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, P.C, s0, 4);
        mn_cc_cls(s_min, s_max, clsmin, clsmax, r.x, s0, lo1, hi1);
      }
    } else {
      // a lane across a component boundary: queued, and worked off below by ALL lanes of the block,
      // one (lane, class) item each -- kept inline here it held registers the streaming lanes never
      // use (76 VGPRs: one block per CU) and its plane reads were nine round trips in a row
      s_q[atomicAdd(&s_qn, 1)] = i;
    }
  }
  }
  __syncthreads();
  {
    const int nq = s_qn;
    for (int t = threadIdx.x; t < nq * P.C; t += MN_CC_SUM_THREADS) {
      const int item = t / P.C, c = t - item * P.C;
      const int g = s_q[item];
      if (clean) {
        const uchar4 cl = *reinterpret_cast<const uchar4*>(clean + 4 * (size_t)g);
        const unsigned char clj[4] = {cl.x, cl.y, cl.z, cl.w};
        const uchar4 bq = *reinterpret_cast<const uchar4*>(cls0 + 4 * (size_t)g);
        const unsigned char bj[4] = {bq.x, bq.y, bq.z, bq.w};
        for (int j = 0; j < 4; j++) {
          const int px = 4 * g + j;
          if (!clj[j]) { if (c == 0) S.osize[px] = 1; continue; }
          const int rt = mn_cc_root_ro(S.parent, px);
          const int sl = mn_lds_root_slot(s_root, rt);
          const float v = mn_ld_class(P, c, px);
          mn_cc_add(P, S, s_root, s_val, lp_acc, rt, c, sl, (i64)__float2int_rn(logf(v) * (float)MN_LP_FIX));
          if (c == 0) {
            mn_cc_add(P, S, s_root, s_val, lp_acc, rt, P.C, sl, 1);
            mn_cc_cls(s_min, s_max, clsmin, clsmax, rt, sl, bj[j], bj[j]);
          }
        }
        continue;
      }
      int4 r;                                             // (a chase from whatever is visible ends at the root)
      r.x = mn_cc_root_ro(S.parent, 4 * g); r.y = mn_cc_root_ro(S.parent, 4 * g + 1);
      r.z = mn_cc_root_ro(S.parent, 4 * g + 2); r.w = mn_cc_root_ro(S.parent, 4 * g + 3);
      const int s0 = mn_lds_root_slot(s_root, r.x), s1 = mn_lds_root_slot(s_root, r.y);
      const int s2 = mn_lds_root_slot(s_root, r.z), s3 = mn_lds_root_slot(s_root, r.w);
      const float4 v = *reinterpret_cast<const float4*>(P.cls + (size_t)c * P.N + 4 * (size_t)g);
      mn_cc_sums_pixelwise(P, S, s_root, s_val, lp_acc, r, c, s0, s1, s2, s3, v);
      if (c == 0) {
        const uchar4 b = *reinterpret_cast<const uchar4*>(cls0 + 4 * (size_t)g);
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, P.C, s0, 1);
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.y, P.C, s1, 1);
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.z, P.C, s2, 1);
        mn_cc_add(P, S, s_root, s_val, lp_acc, r.w, P.C, s3, 1);
        mn_cc_cls(s_min, s_max, clsmin, clsmax, r.x, s0, b.x, b.x);
        mn_cc_cls(s_min, s_max, clsmin, clsmax, r.y, s1, b.y, b.y);
        mn_cc_cls(s_min, s_max, clsmin, clsmax, r.z, s2, b.z, b.z);
        mn_cc_cls(s_min, s_max, clsmin, clsmax, r.w, s3, b.w, b.w);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < MN_CC_SUM_SLOTS && s_root[threadIdx.x] >= 0) {
    atomicMin(&clsmin[s_root[threadIdx.x]], s_min[threadIdx.x]);
    atomicMax(&clsmax[s_root[threadIdx.x]], s_max[threadIdx.x]);
  }
  for (int j = threadIdx.x; j < nval; j += MN_CC_SUM_THREADS) {
    const u64 v = s_val[j];
    if (v == 0) continue;
    const int slot = j / (P.C + 1), c = j - slot * (P.C + 1);
    mn_cc_add(P, S, s_root, s_val, lp_acc, s_root[slot], c, -1, (i64)v);
  }
}

// insert with a bounded probe sequence: the table is sized for "few records between components";
// an input that is not separable may produce millions, so a full table must end the sweep (the
// caller falls back) instead of spinning
__device__ __forceinline__ bool mn_tab_insert_bounded(const HashTab& T, u64 key, i64 s,
                                                      int* __restrict__ tcount = nullptr,
                                                      int count = 0) {
  unsigned slot = mn_hash(key) & T.mask;
#pragma unroll 1
  for (int t = 0; t < 256; t++) {
    const u64 prev = atomicCAS(&T.key[slot], MN_EMPTY, key);
    if (prev == MN_EMPTY || prev == key) {
      atomicAdd(reinterpret_cast<u64*>(&T.S[slot]), (u64)s);
      if (tcount) atomicAdd(&tcount[slot], count);      // pixel edges folded into this record
      T.touched[slot] = 1;
      return true;
    }
    slot = (slot + 1) & T.mask;
  }
  return false;
}

#define MN_CC_EDGE_SLOTS 256
// with the number of pixel edges per record alongside the sum (edge sweep of the fast certificate)
__device__ __forceinline__ bool mn_cc_lds_add_cnt(u64* s_key, u64* s_sum, int* s_cnt, const HashTab& T,
                                                  int* __restrict__ tcount, u64 key, i64 s, int n) {
  unsigned h = (mn_hash(key) >> 7) & (MN_CC_EDGE_SLOTS - 1);
#pragma unroll 1
  for (int t = 0; t < 32; t++) {
    u64 prev = s_key[h];
    if (prev == MN_EMPTY) prev = atomicCAS(&s_key[h], MN_EMPTY, key);
    if (prev == MN_EMPTY || prev == key) {
      atomicAdd(&s_sum[h], (u64)s);
      atomicAdd(&s_cnt[h], n);
      return true;
    }
    h = (h + 1) & (MN_CC_EDGE_SLOTS - 1);
  }
  return mn_tab_insert_bounded(T, key, s, tcount, n);
}


// ---- records between components from the negative out-edges ------------------------------------------
// Every set bit of `negbits` is an in-bounds edge with value <= sep_lo.  A block takes the pixels of one block
// of the sweep: scan of the lanes' bit counts, the edges queued in LDS as 2-byte items ((pixel within the block
// << 5) | offset) and worked off by all lanes -- a lane's own pixels hold 0..40 of them.  Both ends in one
// component: condition (a) fails (a negative edge inside a component).  Otherwise its log-odds (2^-30 fixed
// point: mn_edge_fixed of the value, read again here) go to the record of the two components: block table in
// LDS first -- the entries of a block come from neighbouring pixels and share a handful of records -- then one
// bounded insert per record and block into the global table, with the number of pixel edges alongside
// (certificate).  parent[] is flat here.
#define MN_CC_CROSS_THREADS 256
template <int PX>
__global__ __launch_bounds__(MN_CC_CROSS_THREADS) void mn_cc_cross(
    ImgParams P, const int* __restrict__ parent, HashTab T, const unsigned* __restrict__ negbits,
    int* __restrict__ violations, int* __restrict__ tcount) {
  constexpr int QCAP = MN_CC_CROSS_THREADS * PX * MN_MAX_OFFSETS > 10240 ? 10240 : MN_CC_CROSS_THREADS * PX * MN_MAX_OFFSETS;
  __shared__ unsigned short s_item[QCAP];
  __shared__ int s_w[MN_CC_CROSS_THREADS / 64];
  __shared__ u64 s_key[MN_CC_EDGE_SLOTS];
  __shared__ u64 s_sum[MN_CC_EDGE_SLOTS];
  __shared__ int s_cnt[MN_CC_EDGE_SLOTS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ngroups = (P.N + PX - 1) / PX;
  const int i = blockIdx.x * MN_CC_CROSS_THREADS + threadIdx.x;
  unsigned ng[PX];
#pragma unroll
  for (int j = 0; j < PX; j++) ng[j] = 0u;
  if (i < ngroups) {
    if (PX == 4) {
      const uint4 t = *reinterpret_cast<const uint4*>(negbits + 4 * (size_t)i);
      ng[0] = t.x; ng[1 % PX] = t.y; ng[2 % PX] = t.z; ng[3 % PX] = t.w;
    } else {
      ng[0] = negbits[i];
    }
  }
  int nneg = 0;
#pragma unroll
  for (int j = 0; j < PX; j++) nneg += __popc(ng[j]);
  int incl = nneg;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_w[wave] = incl;
  if (threadIdx.x < MN_CC_EDGE_SLOTS) {
    s_key[threadIdx.x] = MN_EMPTY; s_sum[threadIdx.x] = 0; s_cnt[threadIdx.x] = 0;
  }
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < MN_CC_CROSS_THREADS / 64; w++) { if (w < wave) woff += s_w[w]; total += s_w[w]; }
  if (total == 0) return;                                              // uniform
  int bad = 0, over = 0;
  for (int q0 = 0; q0 < total; q0 += QCAP) {                           // (one pass unless O * PX * 256 > QCAP)
    int pos = woff + incl - nneg - q0;
#pragma unroll
    for (int j = 0; j < PX; j++) {
      unsigned bitsleft = ng[j];
      while (bitsleft) {
        const int k = __ffs((int)bitsleft) - 1;
        bitsleft &= bitsleft - 1u;
        if (pos >= 0 && pos < QCAP) s_item[pos] = (unsigned short)(((threadIdx.x * PX + j) << 5) | k);
        pos++;
      }
    }
    __syncthreads();
    const int nq = min(QCAP, total - q0);
    for (int t0 = 0; t0 < nq; t0 += MN_CC_CROSS_THREADS) {             // (uniform trip count)
      const int t = t0 + threadIdx.x;
      u64 key = MN_EMPTY;
      i64 sx = 0;
      if (t < nq) {
        const unsigned it = s_item[t];
        const int k = (int)(it & 31u);
        const int p = blockIdx.x * (MN_CC_CROSS_THREADS * PX) + (int)(it >> 5);
        const int q = p + P.di[k] * P.W + P.dj[k];
        const float x = mn_same_value(P, P.same[(size_t)k * P.N + p]);
        const int ru = parent[p], rv = parent[q];
        if (ru == rv) bad++;                                           // (a)
        else {
          key = mn_key(ru, rv);
          sx = mn_edge_fixed(x);
        }
      }
      // neighbouring entries come from neighbouring pixels: a wave often holds ONE record, and 64
      // LDS atomics on one address would run one after the other -- reduce in the wave instead
      const u64 k0 = ((u64)(unsigned)__shfl((int)(key >> 32), 0) << 32) | (u64)(unsigned)__shfl((int)(key & 0xFFFFFFFFull), 0);
      if (__ballot(key != k0) == 0) {
        if (k0 == MN_EMPTY) continue;
        i64 tot = sx;
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (lane == 0 && !mn_cc_lds_add_cnt(s_key, s_sum, s_cnt, T, tcount, k0, tot, 64)) over++;
      } else if (key != MN_EMPTY) {
        if (!mn_cc_lds_add_cnt(s_key, s_sum, s_cnt, T, tcount, key, sx, 1)) over++;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < MN_CC_EDGE_SLOTS && s_key[threadIdx.x] != MN_EMPTY)
    if (!mn_tab_insert_bounded(T, s_key[threadIdx.x], (i64)s_sum[threadIdx.x], tcount, s_cnt[threadIdx.x]))
      over++;
  for (int off = 32; off > 0; off >>= 1) { bad += __shfl_xor(bad, off); over += __shfl_xor(over, off); }
  if ((threadIdx.x & 63) == 0) {
    if (bad) atomicAdd(violations, bad);
    if (over) atomicAdd(violations + 1, over);       // table full: not a verdict on the maps
  }
}

// fixed-point sums -> object state; condition (c); the list of component roots (for the
// certificate and the labels, which then need no pass over the pixels).  Only tile roots can be
// component roots: a lane reads the candidate flags of 16 pixels (N bytes in all) and nearly all
// lanes stop there.
__device__ __forceinline__ void mn_cc_finish_root(const ImgParams& P, const ObjState& S, int p,
                                                  const i64* __restrict__ lp_acc,
                                                  const int* __restrict__ clsmin,
                                                  const int* __restrict__ clsmax,
                                                  int* __restrict__ compsize, int* __restrict__ rootlist,
                                                  int* __restrict__ nroots, int* __restrict__ violations,
                                                  const unsigned char* __restrict__ cls0) {
  if (S.parent[p] != p) return;
  if (cls0) {                              // (the sweep left the roots' class and validity flag to us)
    S.ocls[p] = cls0[p];
    S.lpvalid[p] = 0;
  }
  if (rootlist) {                         // (null: cores ahead of the rounds -- no certificate from the contraction)
    rootlist[atomicAdd(nroots, 1)] = p;     // (component roots are few)
    compsize[p] = S.osize[p];               // kept for the certificate: osize grows in the merge
    if (clsmin[p] != clsmax[p]) atomicAdd(violations, 1);                // (c) one class per component
  }
  if (S.osize[p] <= 1) return;            // a lone pixel keeps reading its class planes
  for (int c = 0; c < P.C; c++)
    S.lpsum[(size_t)c * P.N + p] = (float)((double)lp_acc[(size_t)c * P.N + p] * (1.0 / MN_LP_FIX));
  S.lpvalid[p] = 1;
}

__global__ __launch_bounds__(256) void mn_cc_finish(ImgParams P, ObjState S,
                                                    const unsigned char* __restrict__ cand,
                                                    const i64* __restrict__ lp_acc,
                                                    const int* __restrict__ clsmin,
                                                    const int* __restrict__ clsmax,
                                                    int* __restrict__ compsize,
                                                    int* __restrict__ rootlist,
                                                    int* __restrict__ nroots,
                                                    int* __restrict__ violations,
                                                    const unsigned char* __restrict__ cls0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n16 = P.N >> 4;
  if (i < n16) {
    const uint4 f = *reinterpret_cast<const uint4*>(cand + 16 * (size_t)i);
    if ((f.x | f.y | f.z | f.w) != 0u) {
      const unsigned w[4] = {f.x, f.y, f.z, f.w};
      for (int j = 0; j < 16; j++)
        if ((w[j >> 2] >> (8 * (j & 3))) & 0xFFu)
          mn_cc_finish_root(P, S, 16 * i + j, lp_acc, clsmin, clsmax, compsize, rootlist, nroots, violations, cls0);
    }
  }
  if (i < P.N - (n16 << 4)) {
    const int p = (n16 << 4) + i;
    if (cand[p]) mn_cc_finish_root(P, S, p, lp_acc, clsmin, clsmax, compsize, rootlist, nroots, violations, cls0);
  }
}

// ---- certificate and log-likelihood after the merge, without another sweep ----------------------
// total = class term + omf * (sum over edges: log v inside final objects, log(1-v) between them).
// The sign sweep summed both for the components (per block, float64); a record merged afterwards
// moves its edges from "between" to "inside", i.e. adds its log-odds sum, and makes each of them an
// edge whose sign contradicts the partition (the finisher accumulated both).  What is left is per
// COMPONENT: the class term lp[cls] of every final object and the pixels whose own arg-max class
// differs from their final object's class (a component has one class: its size or nothing).  One
// workgroup walks the root list; the class term is summed in 2^-32 fixed point, so the order in
// which the roots were appended does not show in the result.
#define MN_CC_CERT_THREADS 1024
__device__ __forceinline__ void mn_cc_certificate_run(
    const ImgParams& P, const ObjState& S, const unsigned char* __restrict__ cls0,
    const int* __restrict__ compsize, const int* __restrict__ rootlist, int n, int nb_edges,
    const double* __restrict__ partial_edges, const Counters* __restrict__ cnt,
    double* __restrict__ out, int* __restrict__ violations) {
  __shared__ double sh[2][MN_CC_CERT_THREADS];
  __shared__ u64 s_cls;
  __shared__ int s_bad;
  if (threadIdx.x == 0) { s_cls = 0; s_bad = 0; }
  __syncthreads();
  i64 t_cls = 0;
  int bad_cls = 0;
  for (int j = threadIdx.x; j < n; j += MN_CC_CERT_THREADS) {
    const int p = rootlist[j];
    int f = p;
    while (S.parent[f] != f) f = S.parent[f];
    const int oc = S.ocls[f];
    if (cls0[p] != oc) bad_cls += compsize[p];
    if (f == p) t_cls += __float2ll_rn(mn_obj_lp(P, S, S.lpvalid[p] != 0, p, oc) * 4294967296.0f);
  }
  if (t_cls) atomicAdd(&s_cls, (u64)t_cls);
  if (bad_cls) atomicAdd(&s_bad, bad_cls);
  double a0 = 0.0, a1 = 0.0;
  for (int b = threadIdx.x; b < nb_edges; b += MN_CC_CERT_THREADS) {
    a0 += partial_edges[(size_t)b * 2];
    a1 += partial_edges[(size_t)b * 2 + 1];
  }
  sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1;
  __syncthreads();
  for (int off = MN_CC_CERT_THREADS / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double cls_term = (double)(i64)s_cls * (1.0 / 4294967296.0);
    const double moved = (double)cnt->merged_S * (1.0 / MN_FIX_ONE);
    out[0] = cls_term + ((sh[1][0] + sh[0][0]) + moved) * (double)P.omf;
    out[1] = cls_term; out[2] = sh[0][0]; out[3] = sh[1][0];
    if (cnt->merged_E) atomicAdd(violations, cnt->merged_E);
    if (s_bad) atomicAdd(violations + 3, s_bad);
  }
  __syncthreads();
}

__global__ __launch_bounds__(MN_CC_CERT_THREADS) void mn_cc_certificate(
    ImgParams P, ObjState S, const unsigned char* __restrict__ cls0, const int* __restrict__ compsize,
    const int* __restrict__ rootlist, const int* __restrict__ nroots, int nb_edges,
    const double* __restrict__ partial_edges, const Counters* __restrict__ cnt,
    double* __restrict__ out, int* __restrict__ violations) {
  mn_cc_certificate_run(P, S, cls0, compsize, rootlist, *nroots, nb_edges, partial_edges, cnt, out, violations);
}
