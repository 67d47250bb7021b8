// mn_kernels_exact.h -- the reference's sequential order itself, at any image size.
//
// Restates RunSegmentation + Merge (utils/csegment/segment.cc:539-573, 602-727) with the reference's
// own float32 arithmetic (segment.cc:5-46, 107-150): one step = pop the record with the largest
// stored priority, re-score it, merge when the fresh value equals the stored one, otherwise store the
// fresh value; a merge folds / re-keys ONLY the absorbed object's records and re-scores only those
// (:650-707).  A queue entry whose priority differs from the record's is skipped by the reference
// (:554), so its queue is equivalent to "the stored priority of every live record, if >= 0".
//
// What makes a step cost a few global round trips whatever the image size:
//   * queue   = one 32-bit sortable word per record in HBM (`leaf`), blocks of 2^Blog leaves whose
//               maxima (word, record) live in LDS (`l1`, up to 16 K blocks) under a second LDS level
//               (`l2`, one entry per 64 blocks): pop = 4 LDS reads per lane + a wave reduction; an
//               update touches the leaf, the block maximum and its group; a block is re-read from HBM
//               only when its maximum was lowered or removed (the popped block always: read beside
//               the step's other loads)
//   * records = one 16-byte entry per record id = pixel * O + offset index (the reference's creation
//               order, segment.cc:209-231): key (object pair), float32 log-odds sum, table slot
//   * pair -> record: one cuckoo table over (min id, max id) keys, two buckets of four 16-byte slots
//               per key (the reference keeps a hash map per object, segment.h:136): a look-up is two
//               64-byte reads in flight together, whatever the history of the table; a delete frees its
//               slot (no tombstones: the first version's linear probing spent 4 us per merge walking
//               them); the fold look-ups of a merge run in parallel lanes
//   * adjacency = one contiguous array of record ids per object in an arena; a merge walks ONLY the
//               absorbed object's array (64 records per pass) and appends the re-keyed records to the
//               survivor's (doubling reallocation, dead entries dropped on the way).  A single PIXEL's array is
//               IMPLICIT: its records are p * O + k (the pixel as source) and (p - offset k) * O + k (as
//               target) -- nothing is stored, and half of all merges (those that absorb a single pixel) read
//               no adjacency memory at all; an object gets a stored array at its first merge as survivor
// One wavefront runs the loop: no barriers, cross-lane traffic by ballot / shuffle / LDS.  Several
// images can run side by side on different CUs (one context each).
//
// Ties between bit-equal stored priorities go to the lowest record id (the creation order of the
// reference's records); the reference's order among equal keys comes from std::priority_queue's heap
// mechanics, which this does not emulate -- on every reference vector of tests/golden that depends on
// ties (cseg_synth_32x64_n60 among them) the two agree.
#pragma once

#include "mn_device.h"

#define MN_X_INVALID 0xFFFFFFFFu
#define MN_X_HEMPTY 0xFFFFFFFFFFFFFFFFull
#define MN_X_DIRTY 0xFFFFFFFFFFFFFFFFull
#define MN_X_MAXBLOCKS 16384
#define MN_X_TSTACK 1024
#define MN_X_IMPLICIT 0xFFFFFFFFu   /* XObj.aptr of a single pixel: the array is computed, not stored */
// LDS layout of mn_x_run (bytes)
#define MN_X_LDS_GMASK 512
#define MN_X_LDS_CNT (MN_X_LDS_GMASK + 32)
#define MN_X_LDS_TIE (MN_X_LDS_CNT + 32)
#define MN_X_LDS_OFF (MN_X_LDS_TIE + 32)          /* [2 * MN_MAX_OFFSETS] ints */
#define MN_X_LDS_CTAB (MN_X_LDS_OFF + 2 * MN_MAX_OFFSETS * 4)
#define MN_X_LDS_STK (MN_X_LDS_CTAB + 2048 * 4)
#define MN_X_LDS_FREE (MN_X_LDS_STK + MN_X_TSTACK * 8)   /* [MN_X_FREE_CLASSES] heads of the arena's free lists */
#define MN_X_FREE_CLASSES 256
#define MN_X_LDS_L2 (MN_X_LDS_FREE + MN_X_FREE_CLASSES * 4)
#define MN_X_LDS_L1 (MN_X_LDS_L2 + (MN_X_MAXBLOCKS / 64) * 8)
#define MN_X_LDS_BYTES(nbpad) ((size_t)MN_X_LDS_L1 + (size_t)(nbpad) * 8)
// sh_cnt slots
enum { MN_XC_RESCANS = 0, MN_XC_REALLOCS, MN_XC_FOLDED, MN_XC_ADOPTED, MN_XC_SLOW, MN_XC_TIEDMERGES };
// sh_tie slots
enum { MN_XT_DEPTH = 0, MN_XT_PAIRS, MN_XT_TIED, MN_XT_TOPW };

enum { MN_X_RUNNING = 0, MN_X_DONE = 1, MN_X_BUDGET = 2, MN_X_ARENA_FULL = 3, MN_X_HASH_FULL = 4 };

struct XCtl {
  int status;                 // MN_X_*; < 0: mn_status error
  int n_overflow;             // records the parallel set-up could not place in the pair table (negative: placed by the loop's kernel)
  long long steps, merges, rescans, reallocs, folded, adopted, slow_inserts;
  unsigned long long bump;    // next free arena entry
  long long stamps[16];       // -DMN_X_STAMPS (diagnostic build): cycles per phase of the loop
  long long tied_steps;       // pops at which a second live record held the bit-equal stored priority
  long long tied_merges;      // ... of which merged
  long long slot_errors;      // MN_X_CHECK_SLOTS (tests): records / table slots that do not point at each other
  // tie-conflict tracking (see "ties" below): state kept across launches of the loop
  long long tied_conflicts;   // > 0: a choice among bit-equal priorities wrote an object that another such choice read or wrote
  int tdepth, tpairs, ttied;  // nesting stack: entries, adjacent entries with equal words, entries popped while tied
  int ttrack;                 // 1: tracking (stops at the first conflict: the verdict is a yes / no)
};

// one record (AdjacencyRecord, segment.h:175-232): ONE 16-byte load
struct __attribute__((aligned(16))) XRec {
  u64 key;                    // (lower object id << 32) | higher; MN_EMPTY = dead / never existed
  float S;                    // obj_merge_logprob (float32, segment.cc:36, 690)
  unsigned slot;              // slot of the key in the pair table
};

// one slot of the pair table: key, the record that carries it, and a copy of its log-odds sum (a fold
// then needs no further round trip for the sum)
struct __attribute__((aligned(16))) XSlot {
  u64 key;
  unsigned rid;
  float S;
};

// one object (Object, segment.h:85-137)
struct __attribute__((aligned(16))) XObj {
  int size;
  int cls;
  unsigned aptr;              // adjacency array: first arena entry
  int alen;                   //                  entries in use (dead ones included)
};

struct XState {
  XRec* rec;                  // [NL] indexed by record id (pixel * O + k)
  unsigned* leaf;             // [NB << Blog] queue word of the stored priority: 0 = not queued
  XSlot* hs;                  // pair table: nb buckets of 4 slots
  unsigned nb;
  XObj* obj;                  // [N]
  int* acap;                  // [N] adjacency entries owned
  float* lp;                  // [N][C] Object::class_logprobs (float32 sums, segment.cc:640)
  int* parent;                // [N] absorbed -> survivor (the context's union forest)
  unsigned* arena;
  unsigned long long arena_cap;
  unsigned* freeheads;        // [MN_X_FREE_CLASSES] the arena's free lists between launches (heads; a free block's first word links on)
  unsigned* overflow;         // record ids the set-up kernel could not place (both buckets full)
  int overflow_cap;
  u64* l1g;                   // block maxima in HBM (built by mn_x_build_l1, loaded into LDS)
  int Blog, NB, NBpad, NG;
  unsigned NL;                // record ids in use (N * O)
  int* mlog;                  // diagnostic (MN_X_MERGELOG): 4 ints per merge {survivor, absorbed, record, priority bits}
  long long mlog_cap;         // merges the log holds (0: none)
  int dbg;                    // bit 0 (MN_X_FORCE_RELOCATE, tests): a slow insert moves an occupant whenever it can
  unsigned* ostamp;           // [2 N] per object {last event that WROTE its state, last event that read it}: tie-conflict tracking
  u64* tstack;                // [MN_X_TSTACK] the nesting stack between launches
  XCtl* ctl;
};

#include "mn_ref_logf.h"

// differentness_logprob = log(1.0 - same_prob): a double log rounded to float (segment.cc:34)
__device__ __forceinline__ float mn_ref_log1m(float v) { return (float)log(1.0 - (double)v); }

// same_different_bias applied on load (segment.cc:183-195: float logf, double log, expf, double division;
// glibc's logf and expf restated bit for bit: mn_ref_logf.h)
__device__ __forceinline__ float mn_ref_same_value(const ImgParams& P, float v) {
  if (P.clip) v = mn_clip(v);
  if (P.sdb != 0.0f) {
    const float logit = (float)(((double)mn_ref_logf(v) - log(1.0 - (double)v)) + (double)P.sdb);
    v = (float)(1.0 / (1.0 + (double)mn_ref_expf(-logit)));
  }
  return v;
}

__device__ __forceinline__ unsigned mn_x_word(float st) {
  return (st >= 0.0f) ? (((st == 0.0f) ? 0u : __float_as_uint(st)) + 1u) : 0u;
}
__device__ __forceinline__ u64 mn_x_pack(unsigned w, unsigned rid) {
  return w ? (((u64)w << 32) | (u64)(0xFFFFFFFFu - rid)) : 0ull;
}
__device__ __forceinline__ unsigned mn_x_rid(u64 e) { return 0xFFFFFFFFu - (unsigned)e; }

// Wave-wide reductions on the VALU (DPP row moves + four readlanes; a ds_bpermute-based __shfl_xor
// costs an LDS round trip per step, and this loop does several reductions per step).  All lanes get
// the result.
#define MN_X_DPP4(OP, v)                                                                              \
  v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));             \
  v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));             \
  v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));            \
  v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));
__device__ __forceinline__ unsigned mn_x_wmax_u32(unsigned v) {
  MN_X_DPP4(max, v)
  const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}
__device__ __forceinline__ unsigned mn_x_wmin_u32(unsigned v) {
  MN_X_DPP4(min, v)
  const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(r0, r1), min(r2, r3));
}
__device__ __forceinline__ float mn_x_wmax_f32(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false)));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// maximum of (word << 32 | ~record id) entries: the word first, then the low half among its holders
__device__ __forceinline__ u64 mn_x_wmax_pair(u64 v) {
  const unsigned hi = (unsigned)(v >> 32);
  const unsigned mh = mn_x_wmax_u32(hi);
  const unsigned lo = (hi == mh) ? (unsigned)v : 0u;
  const unsigned ml = mn_x_wmax_u32(lo);
  return ((u64)mh << 32) | (u64)ml;
}

// the quotient of the priority (segment.cc:147-149; the Python variant divides by n1 * n2 with the bias
// inside, segmenter.py:190-193)
__device__ __forceinline__ float mn_x_quotient(const ImgParams& P, float num, int n1, int n2) {
  if (P.variant == MN_VARIANT_CSEGMENT)
    return num / (float)((unsigned long long)n1 + (unsigned long long)n2) + P.bias;
  return (num + P.bias) / ((float)n1 * (float)n2);
}

__device__ __forceinline__ u64 mn_hash64(u64 k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}
// the two buckets of a key (any bucket count: multiply-high range reduction of the two hash halves)
__device__ __forceinline__ void mn_x_buckets(u64 key, unsigned nb, unsigned* b1, unsigned* b2) {
  const u64 h = mn_hash64(key);
  const unsigned x = __umulhi((unsigned)h, nb);
  unsigned y = __umulhi((unsigned)(h >> 32), nb);
  if (y == x) y = (x + 1u == nb) ? 0u : x + 1u;
  *b1 = x; *b2 = y;
}

// priority of the record (a, b), a < b by id, by ONE lane (ComputeClassDeltaLogprob +
// UpdateMergePriority, segment.cc:107-150): la / lb = class log-prob vectors of a / b
__device__ __forceinline__ float mn_x_score1(const ImgParams& P, const float* la, const float* lb,
                                             int ca, int cb, int na, int nb, float S, int* mc) {
  float cdl = 0.0f;
  int m = ca;
  if (ca != cb) {
    // sixteen classes of both vectors in flight at once (a loop of dependent loads would pay a
    // memory round trip per class)
    const float lca = la[ca], lcb = lb[cb];
    float bestv = 0.0f;
    m = 0;
    for (int c0 = 0; c0 < P.C; c0 += 16) {
      float va[16], vb[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        va[j] = 0.0f; vb[j] = 0.0f;
        if (c0 + j < P.C) { va[j] = la[c0 + j]; vb[j] = lb[c0 + j]; }
      }
#pragma unroll
      for (int j = 0; j < 16; j++) {
        if (c0 + j < P.C) {
          const float v = va[j] + vb[j];
          if ((c0 + j) == 0 || v > bestv) { bestv = v; m = c0 + j; }
        }
      }
    }
    cdl = (bestv - lca) - lcb;
  }
  *mc = m;
  return mn_x_quotient(P, S * P.omf + cdl, na, nb);
}

// Insertion by ONE lane with everything read afresh (the rare paths: a record the parallel set-up could
// not place, or a lane of a merge pass that lost its slot to another lane): a free slot of either
// bucket, else one occupant is moved to ITS other bucket.  Returns the slot, MN_X_INVALID if no room.
__device__ __noinline__ unsigned mn_x_insert_slow(XSlot* hs, XRec* rec, unsigned nb, u64 key, unsigned rid, float S,
                                                  bool relocate_first = false) {
  unsigned b1, b2;
  mn_x_buckets(key, nb, &b1, &b2);
  XSlot ns; ns.key = key; ns.rid = rid; ns.S = S;
  for (int t = 0; t < 8 && !relocate_first; t++) {
    const unsigned s = (t < 4) ? (b1 * 4 + t) : (b2 * 4 + t - 4);
    if (hs[s].key == MN_X_HEMPTY) { hs[s] = ns; return s; }
  }
  for (int t = 0; t < 8; t++) {
    const unsigned s = (t < 4) ? (b1 * 4 + t) : (b2 * 4 + t - 4);
    const XSlot v = hs[s];
    if (v.key == MN_X_HEMPTY) continue;          // (relocate_first: free slots are taken by the last loop)
    unsigned v1, v2;
    mn_x_buckets(v.key, nb, &v1, &v2);
    const unsigned alt = ((s >> 2) == v1) ? v2 : v1;
    for (int q = 0; q < 4; q++)
      if (hs[alt * 4 + q].key == MN_X_HEMPTY) {
        hs[alt * 4 + q] = v;
        rec[v.rid].slot = alt * 4 + q;
        hs[s] = ns;
        return s;
      }
  }
  for (int t = 0; t < 8 && relocate_first; t++) {
    const unsigned s = (t < 4) ? (b1 * 4 + t) : (b2 * 4 + t - 4);
    if (hs[s].key == MN_X_HEMPTY) { hs[s] = ns; return s; }
  }
  // Two moves: an occupant v of my buckets goes to its other bucket after an occupant w of THAT bucket has gone to
  // its own other bucket.  (With single moves only, one key in ~10^8 insertions found no room at a table load of
  // 0.55 -- twice in eight 1024x2048 images -- and the whole run was repeated with a larger table.)
  for (int t = 0; t < 8; t++) {
    const unsigned s = (t < 4) ? (b1 * 4 + t) : (b2 * 4 + t - 4);
    const XSlot v = hs[s];
    if (v.key == MN_X_HEMPTY) continue;
    unsigned v1, v2;
    mn_x_buckets(v.key, nb, &v1, &v2);
    const unsigned alt = ((s >> 2) == v1) ? v2 : v1;
    if (alt == b1 || alt == b2) continue;
    for (int q = 0; q < 4; q++) {
      const unsigned s2 = alt * 4 + q;
      const XSlot w = hs[s2];
      if (w.key == MN_X_HEMPTY) continue;
      unsigned w1, w2;
      mn_x_buckets(w.key, nb, &w1, &w2);
      const unsigned alt2 = (alt == w1) ? w2 : w1;
      if (alt2 == b1 || alt2 == b2 || alt2 == alt) continue;
      for (int q2 = 0; q2 < 4; q2++)
        if (hs[alt2 * 4 + q2].key == MN_X_HEMPTY) {
          hs[alt2 * 4 + q2] = w;
          rec[w.rid].slot = alt2 * 4 + q2;
          hs[s2] = v;
          rec[v.rid].slot = s2;
          hs[s] = ns;
          return s;
        }
    }
  }
  return MN_X_INVALID;
}

// ---- set-up (parallel, whole chip) -----------------------------------------------------------------
// Object::Object (segment.cc:5-21): lp[c] = logf(p_c), class = first maximum
__global__ __launch_bounds__(256) void mn_x_init_objects(ImgParams P, XState X,
                                                         unsigned char* __restrict__ cls0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  float best = 0.0f;
  int bc = 0;
  for (int c = 0; c < P.C; c++) {
    const float l = mn_ref_logf(mn_ld_class(P, c, p));
    X.lp[(size_t)p * P.C + c] = l;
    if (c == 0 || l > best) { best = l; bc = c; }
  }
  XObj o;
  o.size = 1; o.cls = bc; o.aptr = MN_X_IMPLICIT; o.alen = 2 * P.O;
  X.obj[p] = o;
  X.acap[p] = 0;
  X.parent[p] = p;
  cls0[p] = (unsigned char)bc;
}

// AdjacencyRecord ctor + the constructor's loop (segment.cc:24-46, 209-231): one lane per
// (pixel, offset); records its key, log-odds, initial priority word and table slot (its two adjacency
// entries -- slot k of the source pixel, slot O + k of the target pixel -- are implicit: mn_x_entry)
__global__ __launch_bounds__(256) void mn_x_init_records(ImgParams P, XState X) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)P.N * P.O) return;
  const int p = (int)(gid / P.O), k = (int)(gid - (size_t)p * P.O);
  const unsigned rid = (unsigned)gid;
  const int r = p / P.W, c = p - r * P.W;
  const int rr = r + P.di[k], cc = c + P.dj[k];
  XRec R;
  R.key = MN_EMPTY; R.S = 0.0f; R.slot = MN_X_INVALID;
  if (rr < 0 || rr >= P.H || cc < 0 || cc >= P.W) {
    X.rec[rid] = R;
    X.leaf[rid] = 0u;
    return;
  }
  const int q = rr * P.W + cc;
  const float v = mn_ref_same_value(P, P.same[(size_t)k * P.N + p]);
  const float diff = mn_ref_log1m(v);
  const float same = mn_ref_logf(v);
  const float oml = same - diff;
  const int a = min(p, q), b = max(p, q);
  int mc;
  const float pr = mn_x_score1(P, X.lp + (size_t)a * P.C, X.lp + (size_t)b * P.C, X.obj[a].cls, X.obj[b].cls,
                               1, 1, oml, &mc);
  const u64 key = mn_key(a, b);
  unsigned b1, b2;
  mn_x_buckets(key, X.nb, &b1, &b2);
  // balanced placement: the bucket with more free slots first (the table starts at a load of 0.55; first-fit in
  // the first bucket left every tenth record without a slot at that load, two choices leave a few dozen)
  {
    int f1 = 0, f2 = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      f1 += (X.hs[b1 * 4 + t].key == MN_X_HEMPTY) ? 1 : 0;
      f2 += (X.hs[b2 * 4 + t].key == MN_X_HEMPTY) ? 1 : 0;
    }
    if (f2 > f1) { const unsigned t = b1; b1 = b2; b2 = t; }
  }
  unsigned slot = MN_X_INVALID;
  for (int t = 0; t < 8 && slot == MN_X_INVALID; t++) {
    const unsigned s = (t < 4) ? (b1 * 4 + t) : (b2 * 4 + t - 4);
    if (atomicCAS(&X.hs[s].key, MN_X_HEMPTY, key) == MN_X_HEMPTY) slot = s;
  }
  if (slot != MN_X_INVALID) {
    X.hs[slot].rid = rid;
    X.hs[slot].S = oml;
  } else {
    const int i = atomicAdd(&X.ctl->n_overflow, 1);
    if (i < X.overflow_cap) X.overflow[i] = rid;
  }
  R.key = key; R.S = oml; R.slot = slot;
  X.rec[rid] = R;
  X.leaf[rid] = mn_x_word(pr);
}

// block maxima of the queue words
__global__ __launch_bounds__(64) void mn_x_build_l1(XState X) {
  const unsigned blk = blockIdx.x;
  const unsigned base = blk << X.Blog;
  const int B = 1 << X.Blog;
  u64 m = 0;
  for (int i = threadIdx.x; i < B; i += 64)
    if (base + i < X.NL) {
      const u64 e = mn_x_pack(X.leaf[base + i], base + i);
      m = e > m ? e : m;
    }
  m = mn_x_wmax_pair(m);
  if (threadIdx.x == 0) X.l1g[blk] = m;
}

// ---- the loop ----------------------------------------------------------------------------------------
// One wavefront; cross-lane traffic through LDS is ordered by waiting for the LDS counter (a memory
// clobber keeps the compiler from moving accesses across it).  Global memory: the engine's loads, stores
// and its dependent loads of the same addresses are issued by ONE wave in program order, which the
// memory pipeline preserves per address; -DMN_X_PARANOID drains the vector-memory counter after every
// pass and step (results compared equal with and without on every reference vector).
#define MN_X_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Diagnostic build only (-DMN_X_STAMPS, tests/tools/gpu_exact.py prints them with MN_TRACE_EXACT=1):
// cycles per phase of the loop, every phase ended by a drain of the memory counters so that a phase
// owns its own round trips.  No stamp executes in the product build.
#ifdef MN_X_STAMPS
#define MN_X_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long _t = clock64(); st_acc[k] += _t - st_last; st_last = _t; } while (0)
#else
#define MN_X_STAMP(k) do { } while (0)
#endif
#ifdef MN_X_PARANOID
#define MN_X_MEM_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define MN_X_MEM_SYNC() asm volatile("" ::: "memory")
#endif

// maximum (word, lowest record) of one block of leaves; the whole wave, 16-byte loads, the words first (one
// v_max3 per two leaves), then the lowest id among the holders of the maximum.  The block size is uniform over
// the wave: loads beyond a short block are skipped by scalar branches (no per-lane masks).  The popped record
// is taken out of its block BEFORE the scan by storing 0 in its leaf (the wave's store precedes its loads of
// the same address; the leaf gets its new word -- or stays 0 -- later in the step).
struct __attribute__((packed, aligned(4))) mn_x_f4u { float x, y, z, w; };   // 16-byte load, 4-byte aligned

template <bool WITH_REC>
__device__ __forceinline__ u64 mn_x_scan_block(const unsigned* leaf, unsigned base, int B, int lane,
                                               const XRec* recp, uint4* rec_out) {
  unsigned best = 0u;
  unsigned bestid = MN_X_INVALID;
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  for (int i0 = 0; i0 < B; i0 += 1024) {
    const unsigned* p = leaf + base + (unsigned)(i0 + lane * 4);
    uint4 w0 = *reinterpret_cast<const uint4*>(p), w1 = zero, w2 = zero, w3 = zero;
    if (B > 256) w1 = *reinterpret_cast<const uint4*>(p + 256);
    if (B > 512) { w2 = *reinterpret_cast<const uint4*>(p + 512); w3 = *reinterpret_cast<const uint4*>(p + 768); }
    // (the popped record's own entry: issued BEHIND the leaf loads, so that the wait for the leaves
    //  does not wait for it and the two round trips overlap)
    if (WITH_REC && i0 == 0) *rec_out = *reinterpret_cast<const uint4*>(recp);
    const unsigned m0 = max(max(w0.x, w0.y), max(w0.z, w0.w)), m1 = max(max(w1.x, w1.y), max(w1.z, w1.w));
    const unsigned m2 = max(max(w2.x, w2.y), max(w2.z, w2.w)), m3 = max(max(w3.x, w3.y), max(w3.z, w3.w));
    const unsigned m = max(max(m0, m1), max(m2, m3));
    if (m > best) {                                // (later rounds hold higher ids: ties stay with the earlier)
      best = m;
      // lowest id of this round holding its maximum (descending, so that the lowest wins)
      const unsigned id = base + (unsigned)(i0 + lane * 4);
      unsigned mid = MN_X_INVALID;
      mid = (w3.w == m) ? id + 771 : mid; mid = (w3.z == m) ? id + 770 : mid;
      mid = (w3.y == m) ? id + 769 : mid; mid = (w3.x == m) ? id + 768 : mid;
      mid = (w2.w == m) ? id + 515 : mid; mid = (w2.z == m) ? id + 514 : mid;
      mid = (w2.y == m) ? id + 513 : mid; mid = (w2.x == m) ? id + 512 : mid;
      mid = (w1.w == m) ? id + 259 : mid; mid = (w1.z == m) ? id + 258 : mid;
      mid = (w1.y == m) ? id + 257 : mid; mid = (w1.x == m) ? id + 256 : mid;
      mid = (w0.w == m) ? id + 3 : mid; mid = (w0.z == m) ? id + 2 : mid;
      mid = (w0.y == m) ? id + 1 : mid; mid = (w0.x == m) ? id : mid;
      bestid = mid;
    }
  }
  const unsigned mw = mn_x_wmax_u32(best);
  if (mw == 0u) return 0ull;
  const unsigned mi = mn_x_wmin_u32(best == mw ? bestid : MN_X_INVALID);
  return mn_x_pack(mw, mi);
}

__device__ __forceinline__ void mn_x_group_refresh(u64* l1, u64* l2, int g, int lane) {
  const u64 v = mn_x_wmax_pair(l1[g * 64 + lane]);
  if (lane == 0) l2[g] = v;
}


// ---- ties: does the ORDER among bit-equal priorities matter? -------------------------------------------
// The reference pops bit-equal priorities as its binary heap happens to hold them (segment.h:270-275 compares the
// float only); the engine takes the lowest record id.  The two sequences can only part where a pop had an equal
// rival.  Events (pops) nest: event j hangs under the last earlier event i with word(i) <= word(j) such that every
// event between them is above word(i) -- the suffix minima of the popped words, kept as a stack (words
// non-decreasing, event indices increasing).  Two SIBLINGS with equal words are what a tie rule orders; their
// subtrees (everything popped above their word before the queue falls back to it) commute unless one WRITES the
// state of an object the other reads or writes.  A merge writes the state of its two ends; a pop that only stores
// a fresh priority reads its two ends; a merge reads its third objects (it rewrites their records with the merged
// pair -- every such record has an end the merge writes, and every record an event reads has both ends touched,
// so records need no stamps of their own).  ostamp[2 o] = index of the last event that wrote object o,
// ostamp[2 o + 1] = of the last that read it (not overwritten while that stamp is dangerous, so that a later read
// from the toucher's own subtree cannot hide it).  A stamp s is DANGEROUS for the current event when it lies in
// the subtree of a tied sibling of one of the event's ancestors-or-self: with the stack, "the first entry with
// index > s and the entry before it carry the same word".  Conflict: a read meets a dangerous write stamp, or a
// write meets a dangerous write or read stamp.  Second kind: a merge re-scores or retires a record whose stored word
// equals the word of an entry on the stack that was popped while tied -- a rival whose own turn might have come
// first.  No conflict in a whole run => every order among equals gives the same final state (DESIGN.md
// section 5); tests/tools/exact_model.cpp implements the same criterion on the CPU (XM_RW=1).
// Stack entry: word << 32 | event index << 1 | popped-while-tied.
__device__ __forceinline__ unsigned mn_x_te_word(u64 e) { return (unsigned)(e >> 32); }
__device__ __forceinline__ unsigned mn_x_te_ev(u64 e) { return ((unsigned)e) >> 1; }

// wave-uniform stamp s: conflict?
__device__ __forceinline__ bool mn_x_tie_touch_uniform(const u64* stk, int depth, unsigned s, int lane) {
  if (s == 0u) return false;
  for (int base = 0; base < depth; base += 64) {
    const int j = base + lane;
    const u64 e = (j < depth) ? stk[j] : 0ull;
    const u64 b = __ballot(j < depth && mn_x_te_ev(e) > s);
    if (b) {
      const int i = base + __ffsll((long long)b) - 1;
      if (i == 0) return false;
      return mn_x_te_word(stk[i]) == mn_x_te_word(stk[i - 1]);
    }
  }
  return false;
}
// per-lane stamp s
__device__ __forceinline__ bool mn_x_tie_touch_lane(const u64* stk, int depth, unsigned s) {
  if (s == 0u) return false;
  int lo = 0, hi = depth;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (mn_x_te_ev(stk[mid]) > s) hi = mid; else lo = mid + 1; }
  if (lo >= depth || lo == 0) return false;
  return mn_x_te_word(stk[lo]) == mn_x_te_word(stk[lo - 1]);
}
// per-lane: a record whose stored word w (0 = not queued) is modified: a rival of an event on the stack?
__device__ __forceinline__ bool mn_x_tie_rival_lane(const u64* stk, int depth, unsigned w) {
  if (w == 0u) return false;
  int lo = 0, hi = depth;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (mn_x_te_word(stk[mid]) >= w) hi = mid; else lo = mid + 1; }
  return lo < depth && mn_x_te_word(stk[lo]) == w;
}

// Entry j of an object's adjacency array: stored (arena) or, for a single pixel `o`, computed: j < O the record
// of the pixel as source of offset j (dead -- key MN_EMPTY -- if that edge leaves the image), else the record
// whose TARGET it is under offset j - O.  sh_off: {d_row, d_col} per offset, in LDS.
__device__ __forceinline__ unsigned mn_x_entry(const unsigned* __restrict__ arena, unsigned aptr, int j, int len, int o,
                                              int O, int W, int H, const int* sh_off) {
  if (j >= len) return MN_X_INVALID;
  if (aptr != MN_X_IMPLICIT) return arena[(size_t)aptr + j];
  if (j < O) return (unsigned)o * (unsigned)O + (unsigned)j;
  const int k = j - O;
  const int r = o / W, c = o - r * W;
  const int rr = r - sh_off[2 * k], cc = c - sh_off[2 * k + 1];
  if (rr < 0 || rr >= H || cc < 0 || cc >= W) return MN_X_INVALID;
  return (unsigned)(rr * W + cc) * (unsigned)O + (unsigned)k;
}

// One workgroup (= one wavefront) per image: block b runs the loop of image b of a batch (images are
// independent; the reference scales the same way, by processes).
__global__ __launch_bounds__(64) void mn_x_run(const ImgParams* __restrict__ Ps, const XState* __restrict__ Xs,
                                               long long budget) {
  const ImgParams& P = Ps[blockIdx.x];
  // (the loop is short of scalar registers -- 106 SGPRs with ~100 spills to vector lanes: fields the loop does
  //  not use stay behind the pointer and are loaded where they are needed)
  const XState* __restrict__ Xc = Xs + blockIdx.x;
  struct { XRec* rec; unsigned* leaf; XSlot* hs; unsigned nb; XObj* obj; int* acap; float* lp; int* parent;
           unsigned* arena; unsigned* ostamp; int Blog, NG; XCtl* ctl; } X;
  X.rec = Xc->rec; X.leaf = Xc->leaf; X.hs = Xc->hs; X.nb = Xc->nb; X.obj = Xc->obj; X.acap = Xc->acap;
  X.lp = Xc->lp; X.parent = Xc->parent; X.arena = Xc->arena; X.ostamp = Xc->ostamp; X.Blog = Xc->Blog;
  X.NG = Xc->NG; X.ctl = Xc->ctl;
  {
    // (a relaunch of the batch: this image has finished, or waits for a larger workspace)
    const int st0 = X.ctl->status;
    if (st0 != MN_X_RUNNING && st0 != MN_X_BUDGET) return;
  }
  // LDS layout: the small arrays at FIXED offsets (no scalar registers for their addresses), the block maxima last
  extern __shared__ __attribute__((aligned(16))) unsigned char x_smem[];
  float* sh_lpa = reinterpret_cast<float*>(x_smem);                               // [128] survivor's new class vector
  unsigned* sh_gmask = reinterpret_cast<unsigned*>(x_smem + MN_X_LDS_GMASK);      // [8] groups that lost a maximum
  unsigned* sh_cnt = reinterpret_cast<unsigned*>(x_smem + MN_X_LDS_CNT);          // [8] diagnostic counters of the launch
  int* sh_tie = reinterpret_cast<int*>(x_smem + MN_X_LDS_TIE);                    // [8] tie tracking: depth, pairs, tied entries, top word
  int* sh_off = reinterpret_cast<int*>(x_smem + MN_X_LDS_OFF);                    // [2 * O] offsets (d_row, d_col): implicit adjacency
  unsigned* sh_ctab = reinterpret_cast<unsigned*>(x_smem + MN_X_LDS_CTAB);        // [2048] same-slot check of a pass's inserts
  u64* sh_stk = reinterpret_cast<u64*>(x_smem + MN_X_LDS_STK);                    // [MN_X_TSTACK] nesting stack (ties)
  unsigned* sh_free = reinterpret_cast<unsigned*>(x_smem + MN_X_LDS_FREE);        // [256] first free arena block of 32 * c entries (MN_X_INVALID: none)
  u64* l2 = reinterpret_cast<u64*>(x_smem + MN_X_LDS_L2);                         // [MN_X_MAXBLOCKS / 64] group maxima
  u64* l1 = reinterpret_cast<u64*>(x_smem + MN_X_LDS_L1);                         // [NBpad] block maxima
  const int lane = threadIdx.x;
  const int C = P.C;
  const int B = 1 << X.Blog;

  for (int i = lane; i < Xc->NBpad; i += 64) l1[i] = (i < Xc->NB) ? Xc->l1g[i] : 0ull;
  if (lane < 8) { sh_gmask[lane] = 0u; sh_cnt[lane] = 0u; }
  if (lane < P.O) { sh_off[2 * lane] = P.di[lane]; sh_off[2 * lane + 1] = P.dj[lane]; }
  for (int i = lane; i < MN_X_FREE_CLASSES; i += 64) sh_free[i] = Xc->freeheads[i];
  for (int i = lane; i < MN_X_MAXBLOCKS / 64; i += 64) l2[i] = 0ull;
  MN_X_LDS_SYNC();
  for (int g = 0; g < X.NG; g++) mn_x_group_refresh(l1, l2, g, lane);
  MN_X_LDS_SYNC();

  // (per-launch counts fit 32 bits: a launch runs at most 2^24 steps of at most a few thousand records each)
  unsigned steps = 0, merges = 0, tied_steps = 0;     // (the diagnostic counts live in LDS: sh_cnt)
  const unsigned budget32 = budget > 0x7FFFFFFFll ? 0x7FFFFFFFu : (unsigned)budget;
  // tie-conflict tracking: state of the previous launch
  const long long steps0 = X.ctl->steps;
  const bool ev_overflow = steps0 + (long long)budget32 >= 0x7FFFFFF0ll;
  const unsigned ev0 = (unsigned)steps0 + 1u;
  unsigned tied_conflicts = X.ctl->tied_conflicts > 0 ? 1u : 0u;
  bool track = X.ctl->ttrack != 0;
  {
    // (the nesting stack's scalars live in LDS and are only read while tracking is on: on maps full of equal
    //  values the first conflict comes within a few hundred steps and the loop carries one flag from then on)
    const int d0 = X.ctl->tdepth;
    for (int i = lane; i < d0; i += 64) sh_stk[i] = Xc->tstack[i];
    MN_X_LDS_SYNC();
    if (lane == 0) {
      sh_tie[MN_XT_DEPTH] = d0; sh_tie[MN_XT_PAIRS] = X.ctl->tpairs; sh_tie[MN_XT_TIED] = X.ctl->ttied;
      sh_tie[MN_XT_TOPW] = d0 > 0 ? (int)mn_x_te_word(sh_stk[d0 - 1]) : 0;
    }
    MN_X_LDS_SYNC();
  }
  unsigned long long bump = X.ctl->bump;
  int status = X.ctl->status < 0 || X.ctl->status == MN_X_HASH_FULL ? X.ctl->status : MN_X_RUNNING;
  {
    // The records both of whose buckets were full when the parallel set-up came by: placed here, one by one, by a
    // single lane that may move occupants to their other buckets.  Inside the loop's kernel because the images of
    // a batch then do it side by side (as a kernel of its own it ran once per image, one after the other).
    const int nov = X.ctl->n_overflow;
    if (nov > Xc->overflow_cap) status = MN_X_HASH_FULL;
    else if (nov > 0 && status == MN_X_RUNNING) {
      int bad = 0;
      if (lane == 0) {
        const unsigned* ovf = Xc->overflow;
        for (int i = 0; i < nov && !bad; i++) {
          const unsigned orid = ovf[i];
          const XRec R = X.rec[orid];
          const unsigned sl = mn_x_insert_slow(X.hs, X.rec, X.nb, R.key, orid, R.S);
          if (sl == MN_X_INVALID) bad = 1; else X.rec[orid].slot = sl;
        }
        X.ctl->n_overflow = -nov;               // (done; the count stays readable)
      }
      if (__shfl(bad, 0)) status = MN_X_HASH_FULL;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }

#ifdef MN_X_STAMPS
  long long st_acc[16];
  for (int i = 0; i < 16; i++) st_acc[i] = 0;
  long long st_last = clock64();
#endif

  // Look-ahead: after a step that only refreshes a priority the NEXT pop is almost always the runner-up of this
  // one -- the best entry outside the popped block, or the popped block's new maximum -- which is known as soon as
  // the block has been scanned.  Its record (16 bytes) is fetched while this step waits for its objects; if the
  // next pop is indeed that record, its objects are requested together with its block scan: one round trip
  // instead of two (a step of this kind is 73 % of all steps).  A merge changes records: the guess is dropped.
  unsigned pred_rid = MN_X_INVALID;
  uint4 pred_raw = make_uint4(0u, 0u, 0u, 0u);
  while (status == MN_X_RUNNING) {
    MN_X_STAMP(0);
    // ---- pop: the largest (word, lowest record id) ----
    // (l2 holds MN_X_MAXBLOCKS / 64 = 256 entries, the unused ones 0: four per lane, kept in registers for the
    //  tie test below)
    const u64 q0 = l2[lane], q1 = l2[lane + 64], q2 = l2[lane + 128], q3 = l2[lane + 192];
    u64 top = q0 > q1 ? q0 : q1;
    { const u64 t2 = q2 > q3 ? q2 : q3; top = t2 > top ? t2 : top; }
    top = mn_x_wmax_pair(top);
    const unsigned gword = (unsigned)(top >> 32);
    if (gword == 0u) { status = MN_X_DONE; break; }
    if (top == MN_X_DIRTY) { status = MN_ERR_INTERNAL; break; }
    if (steps >= budget32) { status = MN_X_BUDGET; break; }
    const unsigned rid = mn_x_rid(top);
    const unsigned blk = rid >> X.Blog;
    // Is the pop forced?  A second live record with the bit-equal stored priority (in another group of
    // blocks, another block of this group, or -- below -- this block) means the reference's heap decides
    // between them (segment.h:270-275 compares the float only); the engine takes the lowest record id.
    const int eqg = (((unsigned)(q0 >> 32) == gword) ? 1 : 0) + (((unsigned)(q1 >> 32) == gword) ? 1 : 0) +
                    (((unsigned)(q2 >> 32) == gword) ? 1 : 0) + (((unsigned)(q3 >> 32) == gword) ? 1 : 0);
    const u64 tg = __ballot(eqg > 0);
    bool tied = (tg & (tg - 1ull)) != 0ull || __ballot(eqg > 1) != 0ull;
    const u64 rowv = l1[(blk & ~63u) + lane];          // the maxima of the popped block's group
    {
      const u64 tb = __ballot((unsigned)(rowv >> 32) == gword);
      tied = tied || (tb & (tb - 1ull)) != 0ull;
    }
    MN_X_STAMP(1);
    // the record, and beside it the popped block without it (its maximum changes either way); with the record
    // already here (look-ahead), the objects' state is requested in the same round trip
    const bool ahead = rid == pred_rid;
    uint4 rraw = pred_raw;
    uint4 ox, oy;
    int capx, capy;
    uint2 stx = make_uint2(0u, 0u), sty = make_uint2(0u, 0u);   // {write stamp, read stamp} of both ends
    float ax0 = 0.0f, ay0 = 0.0f, ax1 = 0.0f, ay1 = 0.0f;
#define MN_X_LOAD_OBJECTS(x_, y_) do {                                                                     \
      ox = *reinterpret_cast<const uint4*>(&X.obj[x_]); oy = *reinterpret_cast<const uint4*>(&X.obj[y_]);  \
      capx = X.acap[x_]; capy = X.acap[y_];                                                                \
      if (track) { stx = *reinterpret_cast<const uint2*>(&X.ostamp[2 * (size_t)(x_)]);                     \
                   sty = *reinterpret_cast<const uint2*>(&X.ostamp[2 * (size_t)(y_)]); }                   \
      if (lane < C) { ax0 = X.lp[(size_t)(x_) * C + lane]; ay0 = X.lp[(size_t)(y_) * C + lane]; }          \
      if (lane + 64 < C) { ax1 = X.lp[(size_t)(x_) * C + lane + 64]; ay1 = X.lp[(size_t)(y_) * C + lane + 64]; } \
    } while (0)
    if (ahead) {
      const u64 k0 = ((u64)rraw.y << 32) | (u64)rraw.x;
      MN_X_LOAD_OBJECTS(mn_key_u(k0), mn_key_v(k0));
    }
    if (lane == 0) X.leaf[rid] = 0u;               // (out of its block before the scan; see mn_x_scan_block)
    u64 bm;
    if (ahead) bm = mn_x_scan_block<false>(X.leaf, blk << X.Blog, B, lane, nullptr, nullptr);
    else bm = mn_x_scan_block<true>(X.leaf, blk << X.Blog, B, lane, &X.rec[rid], &rraw);
    const u64 key = ((u64)rraw.y << 32) | (u64)rraw.x;
    const float S = __uint_as_float(rraw.z);
    const unsigned slot_r = rraw.w;
    if (key == MN_EMPTY) { status = MN_ERR_INTERNAL; break; }
    tied = tied || (unsigned)(bm >> 32) == gword;
    tied_steps += tied ? 1 : 0;
    MN_X_STAMP(2);
    const int x = mn_key_u(key), y = mn_key_v(key);
    if (!ahead) MN_X_LOAD_OBJECTS(x, y);
#undef MN_X_LOAD_OBJECTS
    {
      // the guess for the next pop: the best entry outside the popped block, or that block's new maximum
      const int g = (int)(blk >> 6);
      u64 m = (lane == (int)(blk & 63u)) ? 0ull : rowv;
      const bool own = lane == (g & 63);
      { const u64 v = (own && (g >> 6) == 0) ? 0ull : q0; m = v > m ? v : m; }
      { const u64 v = (own && (g >> 6) == 1) ? 0ull : q1; m = v > m ? v : m; }
      { const u64 v = (own && (g >> 6) == 2) ? 0ull : q2; m = v > m ? v : m; }
      { const u64 v = (own && (g >> 6) == 3) ? 0ull : q3; m = v > m ? v : m; }
      m = mn_x_wmax_pair(m);
      if (bm > m) m = bm;
      pred_rid = m ? mn_x_rid(m) : MN_X_INVALID;
      if (pred_rid != MN_X_INVALID) pred_raw = *reinterpret_cast<const uint4*>(&X.rec[pred_rid]);
    }
    // ---- ties: this pop's place in the nesting of events ----
    const unsigned ev = ev0 + steps;
    int tdepth = 0, tpairs = 0, ttied = 0;          // (valid while `track`)
    if (track) {
      tdepth = sh_tie[MN_XT_DEPTH]; tpairs = sh_tie[MN_XT_PAIRS]; ttied = sh_tie[MN_XT_TIED];
      unsigned ttopw = (unsigned)sh_tie[MN_XT_TOPW];
      while (tdepth > 0 && ttopw > gword) {
        const int j = tdepth - 1 - lane;
        const u64 e = (j >= 0) ? sh_stk[j] : 0ull;
        const u64 below = (j >= 1) ? sh_stk[j - 1] : 0ull;
        const bool rem = j >= 0 && mn_x_te_word(e) > gword;
        const int cnt = __popcll(__ballot(rem));
        tpairs -= __popcll(__ballot(rem && j >= 1 && mn_x_te_word(below) == mn_x_te_word(e)));
        ttied -= __popcll(__ballot(rem && (e & 1ull)));
        tdepth -= cnt;
        ttopw = tdepth > 0 ? mn_x_te_word(sh_stk[tdepth - 1]) : 0u;
        if (cnt < 64) break;
      }
      if (tdepth >= MN_X_TSTACK || ev_overflow) {
        tied_conflicts++;            // (cannot be followed any further: counted as a conflict, conservatively)
        track = false;
      } else {
        if (tdepth > 0 && ttopw == gword) tpairs++;
        if (lane == 0) sh_stk[tdepth] = ((u64)gword << 32) | ((u64)ev << 1) | (tied ? 1ull : 0ull);
        tdepth++;
        ttied += tied ? 1 : 0;
        if (lane == 0) {
          sh_tie[MN_XT_DEPTH] = tdepth; sh_tie[MN_XT_PAIRS] = tpairs; sh_tie[MN_XT_TIED] = ttied;
          sh_tie[MN_XT_TOPW] = (int)gword;
        }
        MN_X_LDS_SYNC();
      }
    }
    // ---- re-score (segment.cc:560): both objects' state in one round trip ----
    const int nx = (int)ox.x, ny = (int)oy.x, cx = (int)ox.y, cy = (int)oy.y;
    float cdl = 0.0f;
    int mc = cx;
    if (cx != cy) {
      // first maximum of the joint vector: highest value, lowest class among equals
      const float ninf = -__builtin_huge_valf();
      float j = (lane < C) ? (ax0 + ay0) : ninf;
      bool second = false;
      if (lane + 64 < C) {
        const float j1 = ax1 + ay1;
        if (j1 > j) { j = j1; second = true; }
      }
      const float jm = mn_x_wmax_f32(j);
      const u64 m1 = __ballot(j == jm && !second && lane < C);
      const u64 m2 = __ballot(j == jm && second);
      mc = m1 ? (__ffsll((long long)m1) - 1) : (64 + __ffsll((long long)m2) - 1);
      const int cxs = __builtin_amdgcn_readfirstlane(cx), cys = __builtin_amdgcn_readfirstlane(cy);
      const float lx = cxs < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax0), cxs))
                                : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax1), cxs - 64));
      const float ly = cys < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay0), cys))
                                : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay1), cys - 64));
      cdl = (jm - lx) - ly;
    }
    const float f = mn_x_quotient(P, S * P.omf + cdl, nx, ny);
    const unsigned fw = mn_x_word(f);
    steps++;
    if (track) {
      // a merge WRITES the state of both ends, a pop that only stores a fresh priority READS it: a conflict is a
      // write stamp from a tied sibling's subtree, or -- for a merge -- a read stamp from one.  A read stamp is not
      // overwritten while it is dangerous (a later read from this subtree must not hide it from a write).
      const bool merging = P.variant == MN_VARIANT_CSEGMENT ? (fw == gword) : (fw >= gword);
      bool drx = false, dry = false, conf = false;
      if (tpairs > 0) {
        drx = mn_x_tie_touch_uniform(sh_stk, tdepth, stx.y, lane);
        dry = mn_x_tie_touch_uniform(sh_stk, tdepth, sty.y, lane);
        conf = mn_x_tie_touch_uniform(sh_stk, tdepth, stx.x, lane) || mn_x_tie_touch_uniform(sh_stk, tdepth, sty.x, lane) ||
               (merging && (drx || dry));
      }
      if (lane == 0) {
        if (merging) {
          *reinterpret_cast<uint2*>(&X.ostamp[2 * (size_t)x]) = make_uint2(ev, ev);
          *reinterpret_cast<uint2*>(&X.ostamp[2 * (size_t)y]) = make_uint2(ev, ev);
        } else {
          if (!drx) X.ostamp[2 * (size_t)x + 1] = ev;
          if (!dry) X.ostamp[2 * (size_t)y + 1] = ev;
        }
      }
      if (conf) {
        tied_conflicts++;
        track = false;
      }
    }
    MN_X_STAMP(3);
    // merge when the fresh value is what the queue promised (segment.cc:561; the Python variant
    // merges on >=, segmenter.py:470)
    if (P.variant == MN_VARIANT_CSEGMENT ? (fw != gword) : (fw < gword)) {
      // ---- not what the queue promised: store the fresh value (segment.cc:563-565) ----
      if (lane == 0) {
        X.leaf[rid] = fw;
        const u64 e = mn_x_pack(fw, rid);
        l1[blk] = e > bm ? e : bm;
      }
      MN_X_LDS_SYNC();
      mn_x_group_refresh(l1, l2, (int)(blk >> 6), lane);
      MN_X_LDS_SYNC();
      MN_X_STAMP(4);
      continue;
    }

    // ---- merge (segment.cc:602-727): the larger object survives, a tie keeps the lower id ----
    pred_rid = MN_X_INVALID;                         // (records change below: the look-ahead is dropped)
    const bool swap = nx < ny;
    const int a = swap ? y : x, b = swap ? x : y;
    unsigned pa = swap ? oy.z : ox.z;
    const unsigned pb = swap ? ox.z : oy.z;
    int la = (int)(swap ? oy.w : ox.w);
    const int lb = (int)(swap ? ox.w : oy.w);
    int capa = swap ? capy : capx;
    // room for the records the survivor may adopt; checked before anything is changed, so that a
    // full arena leaves a consistent state behind
    // Blocks of 32 * c entries (c < 256) are recycled through exact-size free lists (heads in LDS, a free
    // block's first word links to the next): the array an absorbed object leaves behind and the one a survivor
    // moves out of serve the next object that reaches that size (64 instead of 86 arena entries per pixel at
    // O = 10, tests/tools/exact_model.cpp).
    bool moved = false, reused = false;
    unsigned newp = 0, free_next = MN_X_INVALID;
    int newcap = 0;
    const unsigned pa_old = pa;
    const int capa_old = capa;
    if (la + lb > capa) {
      newcap = ((2 * (la + lb) + 31) / 32) * 32;    // (doubling)
      const unsigned head = (newcap >> 5) < MN_X_FREE_CLASSES ? sh_free[newcap >> 5] : MN_X_INVALID;
      if (head != MN_X_INVALID) {
        newp = head;
        free_next = X.arena[newp];                   // (read before the move overwrites it; used at the end of the merge)
        reused = true;
      } else {
        if (bump + (unsigned long long)newcap > Xc->arena_cap) { status = MN_X_ARENA_FULL; steps--; break; }
        newp = (unsigned)bump;
        bump += (unsigned long long)newcap;
      }
      moved = true;
    }
    if (Xc->mlog_cap > 0 && lane == 0) {
      const long long at = X.ctl->merges + merges;
      if (at < Xc->mlog_cap) {
        int* ml = Xc->mlog;
        ml[4 * at] = a; ml[4 * at + 1] = b; ml[4 * at + 2] = (int)rid; ml[4 * at + 3] = (int)(gword - 1u);
      }
    }
    merges++;
    if (tied && lane == 0) sh_cnt[MN_XC_TIEDMERGES] += 1u;
    // object state of the survivor (:635-642); the absorbed object only keeps its parent link
    if (lane < C) { const float s0 = ax0 + ay0; X.lp[(size_t)a * C + lane] = s0; sh_lpa[lane] = s0; }
    if (lane + 64 < C) { const float s1 = ax1 + ay1; X.lp[(size_t)a * C + lane + 64] = s1; sh_lpa[lane + 64] = s1; }
    const int na = nx + ny;
    if (lane == 0) {
      X.parent[b] = a;
      X.rec[rid].key = MN_EMPTY;               // the merged record leaves every list (:645-647); its leaf is 0 already
      X.hs[slot_r].key = MN_X_HEMPTY;
      l1[blk] = bm;
      sh_gmask[blk >> 11] |= 1u << ((blk >> 6) & 31u);
    }
    MN_X_LDS_SYNC();
    MN_X_STAMP(5);
    if (moved) {
      // the survivor's live entries move to a larger array
      int cnt = 0;
      for (int j0 = 0; j0 < la; j0 += 64) {
        const int j = j0 + lane;
        const unsigned e = mn_x_entry(X.arena, pa, j, la, a, P.O, P.W, P.H, sh_off);
        bool live = e != MN_X_INVALID && e != rid;
        if (live) live = X.rec[e].key != MN_EMPTY;
        const u64 mk = __ballot(live);
        if (live) X.arena[(size_t)newp + cnt + __popcll(mk & ((1ull << lane) - 1ull))] = e;
        cnt += __popcll(mk);
      }
      pa = newp; la = cnt; capa = newcap;
      if (lane == 0) sh_cnt[MN_XC_REALLOCS] += 1u;
      if (lane == 0) X.acap[a] = capa;
    }
    const float la_c = sh_lpa[mc];
    MN_X_STAMP(6);
    // ---- the absorbed object's records (:650-707), 64 per pass ----
    for (int j0 = 0; j0 < lb; j0 += 64) {
      const int j = j0 + lane;
      const unsigned e = mn_x_entry(X.arena, pb, j, lb, b, P.O, P.W, P.H, sh_off);
      bool live = e != MN_X_INVALID && e != rid;
      u64 kt = MN_EMPTY;
      float St = 0.0f;
      unsigned slot_t = 0;
      unsigned oldw_t = 0u;
      const bool rivals = track && ttied > 0;      // (only then can a touched record be a rival of an event on the stack)
      if (live) {
        const uint4 t = *reinterpret_cast<const uint4*>(&X.rec[e]);
        if (rivals) oldw_t = X.leaf[e];
        kt = ((u64)t.y << 32) | (u64)t.x; St = __uint_as_float(t.z); slot_t = t.w;
      }
      live = live && kt != MN_EMPTY;
      MN_X_STAMP(7);
      const int c3 = live ? ((mn_key_u(kt) == b) ? mn_key_v(kt) : mn_key_u(kt)) : 0;
      // ONE round trip: the third object's state (its first sixteen class terms included) and both
      // buckets of (survivor, third) in the pair table
      int n3 = 0, cc3 = 0;
      u64 key2 = 0;
      int found = -1, freeslot = -1;
      float Su = 0.0f;
      unsigned u_rid = MN_X_INVALID;
      float v3[16];
      uint2 st3 = make_uint2(0u, 0u);
#pragma unroll
      for (int q = 0; q < 16; q++) v3[q] = 0.0f;
      if (live) {
        const float* l3 = X.lp + (size_t)c3 * C;
        const uint4 o3 = *reinterpret_cast<const uint4*>(&X.obj[c3]);
        if (track) st3 = *reinterpret_cast<const uint2*>(&X.ostamp[2 * (size_t)c3]);
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (4 * q < C) {     // (16-byte loads; what lies behind the object's C terms is read and ignored)
            const mn_x_f4u t = *reinterpret_cast<const mn_x_f4u*>(l3 + 4 * q);
            v3[4 * q] = t.x; v3[4 * q + 1] = t.y; v3[4 * q + 2] = t.z; v3[4 * q + 3] = t.w;
          }
        key2 = mn_key(a, c3);
        unsigned b1, b2;
        mn_x_buckets(key2, X.nb, &b1, &b2);
        uint4 sl[8];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          sl[t] = *reinterpret_cast<const uint4*>(&X.hs[b1 * 4 + t]);
          sl[4 + t] = *reinterpret_cast<const uint4*>(&X.hs[b2 * 4 + t]);
        }
        n3 = (int)o3.x; cc3 = (int)o3.y;
#pragma unroll
        for (int t = 7; t >= 0; t--) {
          const u64 hk = ((u64)sl[t].y << 32) | (u64)sl[t].x;
          const int sidx = (int)((t < 4) ? (b1 * 4 + t) : (b2 * 4 + t - 4));
          if (hk == key2) { found = sidx; u_rid = sl[t].z; Su = __uint_as_float(sl[t].w); }
          if (hk == MN_X_HEMPTY) freeslot = sidx;
        }
      }
      MN_X_STAMP(8);
      unsigned tr = MN_X_INVALID;   // the record this lane re-scores
      float Sn = St;
      const bool fold = live && found >= 0;
      const bool adopt = live && found < 0;
      if (track) {
        // a third object's state is READ (its records with the merged pair are rewritten; every such record has an
        // end this merge writes): a conflict only with a write from a tied sibling's subtree
        bool conf = live && tpairs > 0 && mn_x_tie_touch_lane(sh_stk, tdepth, st3.x);
        if (live && !(tpairs > 0 && mn_x_tie_touch_lane(sh_stk, tdepth, st3.y))) X.ostamp[2 * (size_t)c3 + 1] = ev;
        if (rivals) {
          const unsigned oldw_u = fold ? X.leaf[u_rid] : 0u;
          conf = conf || (live && (mn_x_tie_rival_lane(sh_stk, tdepth, oldw_t) || mn_x_tie_rival_lane(sh_stk, tdepth, oldw_u)));
        }
        if (__ballot(conf)) { tied_conflicts++; track = false; }
      }
      if (fold) {
        // the survivor already has a record with the third object: add (:690-692), retire this one (:694)
        tr = u_rid;
        Sn = Su + St;
        X.rec[tr].S = Sn;
        X.hs[found].S = Sn;
        X.rec[e].key = MN_EMPTY;
        X.leaf[e] = 0u;
        X.hs[slot_t].key = MN_X_HEMPTY;
      }
      const u64 am = __ballot(adopt);
      if (am) {
        // re-key (:659-664, 677), adopt into the survivor's list (:700-702).  Two lanes of a pass may
        // have chosen the same free slot (or a lane found no free slot): the first writer of a
        // 2048-entry LDS tag keeps its slot, the others insert one after the other with fresh reads
        const unsigned tag = ((unsigned)freeslot ^ ((unsigned)freeslot >> 11)) & 2047u;
        if (adopt && freeslot >= 0) sh_ctab[tag] = (unsigned)lane;
        MN_X_LDS_SYNC();
        const bool clash = adopt && (freeslot < 0 || sh_ctab[tag] != (unsigned)lane);
        unsigned nslot = (unsigned)freeslot;
        if (adopt) {
          tr = e;
          X.hs[slot_t].key = MN_X_HEMPTY;
          X.arena[(size_t)pa + la + __popcll(am & ((1ull << lane) - 1ull))] = e;
        }
        // Every lane records its slot in X.rec[e] BEFORE a later slow insert can run: a slow insert may move
        // an occupant of a full bucket -- possibly a record placed a moment ago by a lane of this very pass --
        // to its other bucket and then corrects rec[occupant].slot; a record written afterwards would
        // bring the old slot back and a later delete would free another pair's entry.
        if (adopt && !clash) {
          XSlot ns; ns.key = key2; ns.rid = e; ns.S = St;
          *reinterpret_cast<uint4*>(&X.hs[freeslot]) = *reinterpret_cast<const uint4*>(&ns);
          XRec nr; nr.key = key2; nr.S = St; nr.slot = nslot;
          *reinterpret_cast<uint4*>(&X.rec[e]) = *reinterpret_cast<const uint4*>(&nr);
        }
        u64 cm = __ballot(clash);
        while (cm) {
          const int l = __ffsll((long long)cm) - 1;
          cm &= cm - 1ull;
          if (lane == 0) sh_cnt[MN_XC_SLOW] += 1u;
          if (lane == l) {
            nslot = mn_x_insert_slow(X.hs, X.rec, X.nb, key2, e, St, (Xc->dbg & 1) != 0);
            XRec nr; nr.key = key2; nr.S = St; nr.slot = nslot;
            *reinterpret_cast<uint4*>(&X.rec[e]) = *reinterpret_cast<const uint4*>(&nr);
          }
          MN_X_MEM_SYNC();
        }
        if (__ballot(adopt && nslot == MN_X_INVALID)) { status = MN_X_HASH_FULL; break; }
      }
      MN_X_STAMP(9);
      la += __popcll(am);
      {
        const unsigned nfold = (unsigned)__popcll(__ballot(fold));
        if (lane == 0) { sh_cnt[MN_XC_ADOPTED] += (unsigned)__popcll(am); sh_cnt[MN_XC_FOLDED] += nfold; }
      }
      // re-score what was touched (:695-698, 703-706) with the survivor's new state
      unsigned w = 0u;
      if (live) {
        float cdl3 = 0.0f;
        if (mc != cc3) {
          const float* l3 = X.lp + (size_t)c3 * C;
          float l3_c = 0.0f;
          if (cc3 < 16) {
#pragma unroll
            for (int q = 0; q < 16; q++) l3_c = (q == cc3) ? v3[q] : l3_c;
          } else {
            l3_c = l3[cc3];
          }
          float bestv = 0.0f;
#pragma unroll
          for (int q = 0; q < 16; q++)
            if (q < C) { const float v = sh_lpa[q] + v3[q]; if (q == 0 || v > bestv) bestv = v; }
          for (int c0 = 16; c0 < C; c0 += 16) {
            float vb[16];
#pragma unroll
            for (int q = 0; q < 16; q++) vb[q] = (c0 + q < C) ? l3[c0 + q] : 0.0f;
#pragma unroll
            for (int q = 0; q < 16; q++)
              if (c0 + q < C) { const float v = sh_lpa[c0 + q] + vb[q]; if (v > bestv) bestv = v; }
          }
          cdl3 = (a < c3) ? ((bestv - la_c) - l3_c) : ((bestv - l3_c) - la_c);
        }
        w = mn_x_word(mn_x_quotient(P, Sn * P.omf + cdl3, na, n3));
        X.leaf[tr] = w;
      }
      // ---- queue: a raised or new value goes into the block and group maxima by LDS atomics; a block
      //      whose maximum was lowered or removed is marked, and re-read at the end of the step ----
      if (fold) {
        const unsigned be = e >> X.Blog;
        const u64 cur = l1[be];
        if (cur != 0ull && mn_x_rid(cur) == e) {
          atomicMax(&l1[be], MN_X_DIRTY);
          atomicOr(&sh_gmask[be >> 11], 1u << ((be >> 6) & 31u));
        }
      }
      if (live) {
        const unsigned bt = tr >> X.Blog;
        const u64 cur = l1[bt];
        const u64 ne = mn_x_pack(w, tr);
        if (cur != 0ull && cur != MN_X_DIRTY && mn_x_rid(cur) == tr && ne < cur) {
          atomicMax(&l1[bt], MN_X_DIRTY);
          atomicOr(&sh_gmask[bt >> 11], 1u << ((bt >> 6) & 31u));
        } else if (ne) {
          atomicMax(&l1[bt], ne);
          atomicMax(&l2[bt >> 6], ne);
        }
      }
      MN_X_LDS_SYNC();
      MN_X_MEM_SYNC();   // (paranoid build: this pass's stores before the next pass's loads)
      MN_X_STAMP(10);
    }
    if (status != MN_X_RUNNING) break;
    if (lane == 0) {
      XObj o; o.size = na; o.cls = mc; o.aptr = pa; o.alen = la;
      X.obj[a] = o;
      // the arena's free lists: the block taken leaves its list, then the blocks this merge vacated join theirs
      if (reused) sh_free[newcap >> 5] = free_next;
      if (moved && pa_old != MN_X_IMPLICIT && capa_old >= 32 && (capa_old >> 5) < MN_X_FREE_CLASSES && (capa_old & 31) == 0) {
        X.arena[pa_old] = sh_free[capa_old >> 5];
        sh_free[capa_old >> 5] = pa_old;
      }
      const int capb = swap ? capx : capy;
      if (pb != MN_X_IMPLICIT && capb >= 32 && (capb >> 5) < MN_X_FREE_CLASSES && (capb & 31) == 0) {
        X.arena[pb] = sh_free[capb >> 5];
        sh_free[capb >> 5] = pb;
      }
    }
    MN_X_LDS_SYNC();
    MN_X_MEM_SYNC();
    // ---- re-read the blocks whose maximum was lowered or removed, then their groups ----
    for (int wd = 0; wd < 8; wd++) {
      unsigned m = sh_gmask[wd];
      while (m) {
        const int bit = __ffs((int)m) - 1;
        m &= m - 1u;
        const int g = wd * 32 + bit;
        u64 dm = __ballot(l1[g * 64 + lane] == MN_X_DIRTY);
        while (dm) {
          const unsigned bb = (unsigned)(g * 64 + __ffsll((long long)dm) - 1);
          dm &= dm - 1ull;
          const u64 v = mn_x_scan_block<false>(X.leaf, bb << X.Blog, B, lane, nullptr, nullptr);
          if (lane == 0) l1[bb] = v;
          if (lane == 0) sh_cnt[MN_XC_RESCANS] += 1u;
        }
        MN_X_LDS_SYNC();
        mn_x_group_refresh(l1, l2, g, lane);
      }
    }
    MN_X_LDS_SYNC();
    if (lane < 8) sh_gmask[lane] = 0u;
    MN_X_LDS_SYNC();
    MN_X_STAMP(11);
  }

  for (int i = lane; i < sh_tie[MN_XT_DEPTH]; i += 64) Xc->tstack[i] = sh_stk[i];
  for (int i = lane; i < MN_X_FREE_CLASSES; i += 64) Xc->freeheads[i] = sh_free[i];
  if (lane == 0) {
    XCtl* c = X.ctl;
#ifdef MN_X_STAMPS
    for (int i = 0; i < 16; i++) c->stamps[i] += st_acc[i];
#endif
    c->status = status;
    c->steps += steps; c->merges += merges; c->rescans += sh_cnt[MN_XC_RESCANS]; c->reallocs += sh_cnt[MN_XC_REALLOCS];
    c->folded += sh_cnt[MN_XC_FOLDED]; c->adopted += sh_cnt[MN_XC_ADOPTED]; c->slow_inserts += sh_cnt[MN_XC_SLOW];
    c->tied_steps += tied_steps; c->tied_merges += sh_cnt[MN_XC_TIEDMERGES];
    c->tied_conflicts = (long long)tied_conflicts;
    c->tdepth = sh_tie[MN_XT_DEPTH]; c->tpairs = sh_tie[MN_XT_PAIRS]; c->ttied = sh_tie[MN_XT_TIED];
    c->ttrack = track ? 1 : 0;
    c->bump = bump;
  }
}

// Tests (MN_X_CHECK_SLOTS): every live record's slot holds its key and id, every occupied slot names a live
// record with that key that points back at it (a stale slot index would free another pair's entry later).
__global__ __launch_bounds__(256) void mn_x_check_slots(XState X) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int bad = 0;
  if (i < X.NL) {
    const XRec R = X.rec[i];
    if (R.key != MN_EMPTY) {
      if (R.slot == MN_X_INVALID || R.slot >= X.nb * 4u) bad++;
      else { const XSlot s = X.hs[R.slot]; if (s.key != R.key || s.rid != (unsigned)i || s.S != R.S) bad++; }
    }
  }
  if (i < (size_t)X.nb * 4) {
    const XSlot s = X.hs[i];
    if (s.key != MN_X_HEMPTY) {
      if (s.rid >= X.NL) bad++;
      else { const XRec R = X.rec[s.rid]; if (R.key != s.key || R.slot != (unsigned)i) bad++; }
    }
  }
  if (bad) atomicAdd((unsigned long long*)&X.ctl->slot_errors, (unsigned long long)bad);
}

// Phase A as the engine holds it, in the layout of the oracle's phase-A export (tests): per (offset, source
// pixel) the record's log-odds and initial priority, NaN where the edge leaves the image.
__global__ __launch_bounds__(256) void mn_x_export_phase_a(ImgParams P, XState X, float* __restrict__ oml_out,
                                                           float* __restrict__ prio_out,
                                                           unsigned char* __restrict__ cls_out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)P.N * P.O) return;
  const int p = (int)(gid / P.O), k = (int)(gid - (size_t)p * P.O);
  if (k == 0 && cls_out) cls_out[p] = (unsigned char)X.obj[p].cls;
  const XRec R = X.rec[gid];
  const size_t e = (size_t)k * P.N + p;
  if (R.key == MN_EMPTY) {
    oml_out[e] = __int_as_float(0x7FC00000);
    prio_out[e] = __int_as_float(0x7FC00000);
    return;
  }
  const int a = mn_key_u(R.key), b = mn_key_v(R.key);
  int mc;
  oml_out[e] = R.S;
  prio_out[e] = mn_x_score1(P, X.lp + (size_t)a * P.C, X.lp + (size_t)b * P.C, X.obj[a].cls, X.obj[b].cls, 1, 1, R.S, &mc);
}

// ---- hand-over to the output stage -------------------------------------------------------------------
// The output kernels read object sizes, classes and the class sums of live objects (plane-major table,
// mn_obj_lp) from the context's own arrays: copy the survivors' state there.
__global__ __launch_bounds__(256) void mn_x_export_objects(ImgParams P, XState X, int* __restrict__ osize,
                                                           unsigned char* __restrict__ ocls,
                                                           float* __restrict__ lpsum,
                                                           unsigned char* __restrict__ lpvalid) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  const XObj o = X.obj[p];
  osize[p] = o.size;
  ocls[p] = (unsigned char)o.cls;
  const bool root = X.parent[p] == p;
  lpvalid[p] = root ? 1 : 0;
  if (!root) return;
  for (int c = 0; c < P.C; c++) lpsum[(size_t)c * P.N + p] = X.lp[(size_t)p * P.C + c];
}

// Quotient condition of the certificate on the engine's own records: no record between two final
// objects may still be mergeable (cf. mn_verify_records).
__global__ __launch_bounds__(256) void mn_x_verify_records(ImgParams P, XState X, int* __restrict__ violations) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= X.NL) return;
  const XRec R = X.rec[i];
  if (R.key == MN_EMPTY) return;
  const int a = mn_key_u(R.key), b = mn_key_v(R.key);
  const XObj oa = X.obj[a], ob = X.obj[b];
  int mc;
  const float f = mn_x_score1(P, X.lp + (size_t)a * P.C, X.lp + (size_t)b * P.C, oa.cls, ob.cls,
                              oa.size, ob.size, R.S, &mc);
  const float margin = 1e-6f + 1e-5f * fabsf(P.bias);
  if (!(f < -margin)) atomicAdd(violations + 4, 1);
}
