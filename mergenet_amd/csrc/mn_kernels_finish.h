// mn_kernels_finish.h -- sequential lazy-greedy merge in ONE workgroup (exact order).
//
// Restates the reference loop itself (utils/csegment/segment.cc:539-573 with Merge :602-727;
// Python variant utils/segmenter.py:455-473, 485-578) for record lists small enough that a
// block-wide arg-max replaces the priority queue:
//   step:  r = live record with the largest stored priority >= 0
//          (a queue entry whose priority differs from the record's is skipped by the reference,
//           segment.cc:554, so the queue is equivalent to "the stored priority of each record")
//          fresh = re-score(r)                                        (:560)
//          fresh == stored -> merge(r); else stored = fresh           (:561-565)
//   merge: survivor a = larger object (tie: lower id), absorbed b     (:612-616)
//          every record (b,o): fold into (a,o) if it exists, else re-key to (a,o);
//          ONLY these records are re-scored                           (:650-707)
// Ties between equal stored priorities are broken by (lowest u, lowest v); the reference breaks
// them by heap mechanics and hash-map iteration order, which is the one documented difference.
//
// Used (a) as the finisher once the parallel rounds have shrunk the record list, where a round
// per merge would be launch-bound, and (b) as the exact mode for small images.
// Every wave reaches the loop exit: the step counter is bounded by max_steps.
#pragma once

#include "mn_device.h"
#include "mn_kernels_merge.h"

#define MN_FIN_THREADS 1024
#define MN_FIN_WAVES (MN_FIN_THREADS / 64)

struct FinCand {
  unsigned bits;   // stored priority as ordered bits (0 for +-0)
  int u, v, idx;
};

__device__ __forceinline__ bool mn_cand_better(const FinCand& a, const FinCand& b) {
  // true when a should be popped before b; idx < 0 = empty
  if (b.idx < 0) return a.idx >= 0;
  if (a.idx < 0) return false;
  if (a.bits != b.bits) return a.bits > b.bits;
  if (a.u != b.u) return a.u < b.u;
  return a.v < b.v;
}

__global__ __launch_bounds__(MN_FIN_THREADS) void mn_finisher(ImgParams P, ObjState S, RecList L,
                                                              int R, int* __restrict__ mapbuf,
                                                              int* __restrict__ touched_list,
                                                              Counters* __restrict__ cnt,
                                                              long long max_steps) {
  __shared__ FinCand sh_c[MN_FIN_WAVES];
  __shared__ FinCand sh_best;
  __shared__ int sh_do_merge, sh_a, sh_b, sh_ntouched;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  long long steps = 0;
  int merges = 0;

  for (;;) {
    // ---- 1. arg-max of the stored priorities ----
    FinCand best;
    best.idx = -1; best.bits = 0; best.u = 0; best.v = 0;
    for (int i = tid; i < R; i += MN_FIN_THREADS) {
      const u64 key = L.key[i];
      if (key == MN_EMPTY) continue;
      const float st = L.st[i];
      if (!(st >= 0.0f)) continue;
      FinCand c;
      c.bits = (st == 0.0f) ? 0u : __float_as_uint(st);
      c.u = mn_key_u(key); c.v = mn_key_v(key); c.idx = i;
      if (mn_cand_better(c, best)) best = c;
    }
    for (int off = 32; off > 0; off >>= 1) {
      FinCand o;
      o.bits = __shfl_xor(best.bits, off);
      o.u = __shfl_xor(best.u, off);
      o.v = __shfl_xor(best.v, off);
      o.idx = __shfl_xor(best.idx, off);
      if (mn_cand_better(o, best)) best = o;
    }
    if (lane == 0) sh_c[wave] = best;
    __syncthreads();
    if (tid == 0) {
      FinCand b = sh_c[0];
      for (int w = 1; w < MN_FIN_WAVES; w++)
        if (mn_cand_better(sh_c[w], b)) b = sh_c[w];
      sh_best = b;
      sh_do_merge = 0;
      sh_ntouched = 0;
      if (b.idx >= 0) {
        // ---- 2. re-score the popped record ----
        int mc;
        bool pos;
        const float st = L.st[b.idx];
        const float f = mn_score(P, S, b.u, b.v, mn_fixed_to_float(L.S[b.idx]), &mc, &pos);
        const bool go = (P.variant == MN_VARIANT_CSEGMENT) ? (f == st) : (f >= st);
        if (!go) {
          L.st[b.idx] = f;
        } else {
          int a = b.u, bb = b.v;
          if (S.osize[a] < S.osize[bb]) { const int t = a; a = bb; bb = t; }
          sh_a = a; sh_b = bb;
          sh_do_merge = 1 + mc;       // merged class travels with the flag
        }
      }
    }
    __syncthreads();
    if (sh_best.idx < 0) break;
    steps++;
    if (steps > max_steps) { if (tid == 0) cnt->error = MN_ERR_INTERNAL; break; }
    if (!sh_do_merge) continue;

    // ---- 3. merge: object state ----
    const int a = sh_a, b = sh_b;
    const int mcls = sh_do_merge - 1;
    if (tid < P.C) {
      const bool va = S.lpvalid[a] != 0, vb = S.lpvalid[b] != 0;
      const float s = mn_obj_lp(P, S, va, a, tid) + mn_obj_lp(P, S, vb, b, tid);
      S.lpsum[(size_t)tid * P.N + a] = s;
    }
    if (tid == 0) L.key[sh_best.idx] = MN_EMPTY;
    // pass A: where does the survivor already have a record?  mapbuf[o] = index of (a,o)
    for (int i = tid; i < R; i += MN_FIN_THREADS) {
      const u64 key = L.key[i];
      if (key == MN_EMPTY || i == sh_best.idx) continue;
      const int u = mn_key_u(key), v = mn_key_v(key);
      if (u == a) mapbuf[v] = i;
      else if (v == a) mapbuf[u] = i;
    }
    __syncthreads();
    if (tid == 0) {
      S.lpvalid[a] = 1;
      S.ocls[a] = (unsigned char)mcls;
      S.osize[a] = S.osize[a] + S.osize[b];
      S.parent[b] = a;
    }
    // pass B: records of the absorbed object fold into / move to the survivor
    for (int i = tid; i < R; i += MN_FIN_THREADS) {
      const u64 key = L.key[i];
      if (key == MN_EMPTY) continue;
      const int u = mn_key_u(key), v = mn_key_v(key);
      int o;
      if (u == b) o = v; else if (v == b) o = u; else continue;
      const int j = mapbuf[o];
      int t;
      if (j >= 0) {
        L.S[j] += L.S[i];            // one (b,o) per o: no two lanes add to the same j
        L.key[i] = MN_EMPTY;
        t = j;
      } else {
        L.key[i] = mn_key(a, o);
        t = i;
      }
      touched_list[atomicAdd(&sh_ntouched, 1)] = t;
    }
    __syncthreads();
    // pass C: clear the map, re-score ONLY the touched records with the merged object's state
    for (int i = tid; i < R; i += MN_FIN_THREADS) {
      const u64 key = L.key[i];
      if (key == MN_EMPTY) continue;
      const int u = mn_key_u(key), v = mn_key_v(key);
      if (u == a) mapbuf[v] = -1;
      else if (v == a) mapbuf[u] = -1;
    }
    const int nt = sh_ntouched;
    for (int k = tid; k < nt; k += MN_FIN_THREADS) {
      const int t = touched_list[k];
      const u64 key = L.key[t];
      int mc;
      bool pos;
      L.st[t] = mn_score(P, S, mn_key_u(key), mn_key_v(key), mn_fixed_to_float(L.S[t]), &mc, &pos);
    }
    merges++;
    __syncthreads();
  }
  if (tid == 0) {
    cnt->finisher_steps = (int)(steps > 0x7FFFFFFF ? 0x7FFFFFFF : steps);
    cnt->finisher_merges = merges;
    cnt->n_merged = merges;
  }
}
