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
    // pass C: clear the map, re-score ONLY the touched records with the merged object's state.
    // (Lowering the survivor's other, now stale-high, records right away would save pops but is
    // NOT order-neutral: their priority can rise again before the queue reaches them.)
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


// ---- the same loop with the record list resident in LDS ----------------------------------------
// For lists of at most MN_FIN2_MAXR records the keys (pixel-id pairs) and the stored priorities
// live in LDS (96 KiB of the CU's 160 KiB), so the arg-max and the incidence scan of every step
// are LDS scans; global memory is touched only for the object state of the popped record and
// for the records of the two merging objects.  Identical semantics and tie rule as mn_finisher.
// What a step costs is latency, not bandwidth (one workgroup, dependent round trips), hence:
// bits-only max reduction with the tie rule applied only when two records share the maximum,
// wave-aggregated list appends, and no id translation between the scan and the object state.
#define MN_FIN2_MAXR 8192

// Diagnostic build only (-DMN_FIN_STAMPS, tools/fin_stamps.sh): per-phase cycle totals of the
// step loop, printed by lane 0 when the kernel ends.  No stamp executes in the product build.
#ifdef MN_FIN_STAMPS
#define MN_STAMP(k) do { if (tid == 0) { const long long _t = clock64(); st_acc[k] += _t - st_last; st_last = _t; } } while (0)
#else
#define MN_STAMP(k) do { } while (0)
#endif

// append `value` to a list for every lane with `flag`; one LDS atomic per wave
__device__ __forceinline__ void mn_wave_append(bool flag, int value, int* __restrict__ list,
                                               int* counter) {
  const u64 m = __ballot(flag);
  if (m == 0) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (flag) list[base + __popcll(m & ((1ull << lane) - 1ull))] = value;
}

#ifndef MN_FIN2_THREADS
#define MN_FIN2_THREADS 1024   /* measured: 256 -> 117 ms, 512 -> 102 ms, 1024 -> 95 ms per image */
#endif
#define MN_FIN2_WAVES (MN_FIN2_THREADS / 64)

// stored priority <-> sortable LDS word: 0 = not in the queue (priority < 0), else bits + 1
__device__ __forceinline__ unsigned mn_fin_word(float st) {
  return (st >= 0.0f) ? (((st == 0.0f) ? 0u : __float_as_uint(st)) + 1u) : 0u;
}

// The loop itself, callable from a kernel of MN_FIN2_THREADS lanes with MN_FIN2_MAXR * 12 bytes of
// dynamic shared memory (mn_finisher_lds below; mn_cc_tail runs it between its other stages).
__device__ __forceinline__ void mn_fin_lds_run(
    const ImgParams& P, const ObjState& S, const RecList& L, int R, int* __restrict__ maprec,
    int* __restrict__ lists, Counters* __restrict__ cnt, long long max_steps, int* __restrict__ lcount) {
  // One CU runs this loop, and what bounds a step is instruction issue (16 waves share 4 SIMDs),
  // so the per-record work of the two scans is kept to a couple of instructions: the queue is an
  // array of sortable words, the keys are two u32 arrays.
  extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
  unsigned* lw = reinterpret_cast<unsigned*>(fin_smem);                       // [MAXR] queue word
  unsigned* lu = reinterpret_cast<unsigned*>(fin_smem + MN_FIN2_MAXR * 4);    // [MAXR] lower id
  unsigned* lv = reinterpret_cast<unsigned*>(fin_smem + MN_FIN2_MAXR * 8);    // [MAXR] higher id
  __shared__ unsigned sh_bits[MN_FIN2_WAVES];
  __shared__ u64 sh_tie;
  __shared__ int sh_widx, sh_cnt;
  __shared__ int sh_do_merge, sh_a, sh_b, sh_nA, sh_nB, sh_nt, sh_valid;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned DEAD = 0xFFFFFFFFu;
  int* listA = lists;
  int* listB = lists + MN_FIN2_MAXR;
  int* touched = lists + 2 * MN_FIN2_MAXR;

  // the list never grows, so every scan stops at R rounded up to the block size
  const int Rp = min(MN_FIN2_MAXR, (R + MN_FIN2_THREADS - 1) / MN_FIN2_THREADS * MN_FIN2_THREADS);
  for (int i = tid; i < Rp; i += MN_FIN2_THREADS) {
    unsigned w = 0, u = DEAD, v = DEAD;
    if (i < R) {
      const u64 k = L.key[i];
      if (k != MN_EMPTY) {
        u = (unsigned)mn_key_u(k); v = (unsigned)mn_key_v(k); w = mn_fin_word(L.st[i]);
        maprec[u] = -1; maprec[v] = -1;     // the (object -> record) map is only ever read at
      }                                     // endpoints of records: no N-sized clear beforehand
    }
    lw[i] = w; lu[i] = u; lv[i] = v;
  }
  if (tid == 0) { sh_tie = MN_EMPTY; sh_cnt = 0; }
  __syncthreads();

  long long steps = 0;
  int merges = 0;
  // `lcount` (components mode): pixel edges per record; what the merged records held is handed
  // to mn_cc_cert_reduce (lane 0 keeps the totals)
  i64 merged_S = 0;
  int merged_E = 0;
#ifdef MN_FIN_STAMPS
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long st_last = clock64();
#endif
  for (;;) {
    MN_STAMP(7);
    // ---- 1. arg-max: every lane keeps (max word, its index, how many of its entries share it);
    //         the block reduces the word; if exactly one entry holds the maximum -- the common
    //         case -- its lane publishes the index, otherwise the tie rule (lowest u, v) runs ----
    unsigned m = 0;
    int midx = 0, mcnt = 0;
#pragma unroll 4
    for (int i = tid; i < Rp; i += MN_FIN2_THREADS) {
      const unsigned w = lw[i];
      mcnt = (w > m) ? 1 : (mcnt + ((w == m) ? 1 : 0));
      midx = (w > m) ? i : midx;
      m = max(m, w);
    }
    unsigned wm = m;
    for (int off = 32; off > 0; off >>= 1) wm = max(wm, (unsigned)__shfl_xor((int)wm, off));
    if (lane == 0) sh_bits[wave] = wm;
    __syncthreads();
    unsigned gmax = 0;
#pragma unroll
    for (int w = 0; w < MN_FIN2_WAVES; w++) gmax = max(gmax, sh_bits[w]);
    if (gmax == 0) break;                           // queue empty (uniform)
    if (m == gmax) { atomicAdd(&sh_cnt, mcnt); sh_widx = midx; }
    __syncthreads();
    if (sh_cnt > 1) {                               // equal priorities: lowest (u, v) wins
#ifdef MN_FIN_STAMPS
      if (tid == 0) st_acc[6]++;
#endif
      if (wm == gmax) {
        for (int i = tid; i < Rp; i += MN_FIN2_THREADS)
          if (lw[i] == gmax) atomicMin(&sh_tie, ((u64)lu[i] << 32) | (u64)lv[i]);
      }
      __syncthreads();
      const u64 want = sh_tie;
      if (wm == gmax) {
        for (int i = tid; i < Rp; i += MN_FIN2_THREADS)
          if (lw[i] == gmax && (((u64)lu[i] << 32) | (u64)lv[i]) == want) sh_widx = i;
      }
      __syncthreads();
    }
    MN_STAMP(0);
    if (wave == 0) {
      // ---- 2. re-score the popped record: the whole first wave, one global round trip ----
      const int bi = sh_widx;
      const int pu = (int)lu[bi], pv = (int)lv[bi];
      int mc;
      bool pos;
      int fld[4];
      const float f = mn_score_wave(P, S, pu, pv, mn_fixed_to_float(L.S[bi]), &mc, &pos, fld);
      if (tid == 0) {
      sh_tie = MN_EMPTY;
      sh_cnt = 0;
      sh_do_merge = 0;
      sh_nA = 0; sh_nB = 0; sh_nt = 0;
      const unsigned fw = mn_fin_word(f);
      // stored == fresh  <=>  equal queue words (both >= 0); the Python variant merges on >=
      const bool go = (P.variant == MN_VARIANT_CSEGMENT) ? (fw == gmax) : (fw >= gmax);
      if (!go) {
        lw[bi] = fw;
      } else {
        // survivor = larger object, tie keeps the lower id (pu); sizes and validity flags are
        // the ones the score just read
        const bool swap = fld[0] < fld[1];
        sh_a = swap ? pv : pu;
        sh_b = swap ? pu : pv;
        sh_valid = swap ? ((fld[3] ? 1 : 0) | (fld[2] ? 2 : 0)) : ((fld[2] ? 1 : 0) | (fld[3] ? 2 : 0));
        sh_do_merge = 1 + mc;
        lw[bi] = 0; lu[bi] = DEAD; lv[bi] = DEAD;
        if (lcount) { merged_S += L.S[bi]; merged_E += lcount[bi]; }
      }
      }
    }
    __syncthreads();
    MN_STAMP(1);
    steps++;
    if (steps > max_steps) { if (tid == 0) cnt->error = MN_ERR_INTERNAL; break; }
    if (!sh_do_merge) continue;

    // ---- 3. merge ----
    const unsigned a = (unsigned)sh_a, b = (unsigned)sh_b;
    const int mcls = sh_do_merge - 1;
    // one LDS scan: which records touch the survivor, which the absorbed object
    for (int i0 = 0; i0 < Rp; i0 += MN_FIN2_THREADS) {
      const int i = i0 + tid;
      const unsigned u = lu[i], v = lv[i];
      const bool isA = u == a || v == a;
      const bool isB = !isA && (u == b || v == b);
      if (__ballot(isA || isB) == 0) continue;      // wave-uniform: most waves see neither object
      mn_wave_append(isA, i, listA, &sh_nA);
      mn_wave_append(isB, i, listB, &sh_nB);
    }
    __syncthreads();
    MN_STAMP(2);
    const int nA = sh_nA, nB = sh_nB;
    // class log-prob vectors are added by the LAST lanes while the first ones fill the map (both
    // are global round trips); validity flags were read by lane 0 before it rewrites them
    if (tid >= MN_FIN2_THREADS - 128) {
      const bool va = (sh_valid & 1) != 0, vb = (sh_valid & 2) != 0;
      for (int c = tid - (MN_FIN2_THREADS - 128); c < P.C; c += 128)
        S.lpsum[(size_t)c * P.N + a] = mn_obj_lp(P, S, va, (int)a, c) + mn_obj_lp(P, S, vb, (int)b, c);
    }
    for (int j = tid; j < nA; j += MN_FIN2_THREADS) {
      const int i = listA[j];
      maprec[lu[i] == a ? lv[i] : lu[i]] = i;
    }
    if (tid == 0) {
      S.lpvalid[a] = 1;
      S.ocls[a] = (unsigned char)mcls;
      S.osize[a] = S.osize[a] + S.osize[b];
      S.parent[b] = (int)a;
    }
    __syncthreads();
    MN_STAMP(3);
    for (int j0 = 0; j0 < nB; j0 += MN_FIN2_THREADS) {
      const int j = j0 + tid;
      int t = -1;
      if (j < nB) {
        const int i = listB[j];
        const unsigned o = lu[i] == b ? lv[i] : lu[i];
        const int mrec = maprec[o];
        if (mrec >= 0) {
          L.S[mrec] += L.S[i];
          if (lcount) lcount[mrec] += lcount[i];
          lw[i] = 0; lu[i] = DEAD; lv[i] = DEAD;
          t = mrec;
        } else {
          lu[i] = min(a, o); lv[i] = max(a, o);
          t = i;
        }
      }
      mn_wave_append(t >= 0, t, touched, &sh_nt);
    }
    __syncthreads();
    MN_STAMP(4);
    for (int j = tid; j < nA; j += MN_FIN2_THREADS) {
      const int i = listA[j];
      maprec[lu[i] == a ? lv[i] : lu[i]] = -1;
    }
    const int nt = sh_nt;
    if (nt <= 4 * MN_FIN2_WAVES) {
      // few records to re-score: a wave each (one round trip per record instead of three)
      for (int j = wave; j < nt; j += MN_FIN2_WAVES) {
        const int t = touched[j];
        int mc;
        bool pos;
        const float f = mn_score_wave(P, S, (int)lu[t], (int)lv[t], mn_fixed_to_float(L.S[t]), &mc, &pos);
        if (lane == 0) lw[t] = mn_fin_word(f);
      }
    } else {
      for (int j = tid; j < nt; j += MN_FIN2_THREADS) {
        const int t = touched[j];
        int mc;
        bool pos;
        lw[t] = mn_fin_word(mn_score(P, S, (int)lu[t], (int)lv[t], mn_fixed_to_float(L.S[t]), &mc, &pos));
      }
    }
    merges++;
    __syncthreads();
    MN_STAMP(5);
  }
#ifdef MN_FIN_STAMPS
  if (tid == 0)
    printf("fin stamps (cycles): argmax %lld decide %lld lists %lld map %lld fold %lld rescore %lld loop %lld | steps %lld merges %d ties %lld\n",
           st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5], st_acc[7], steps, merges, st_acc[6]);
#endif
  // ---- epilogue: records back to the global list (dead ones marked; priorities are not needed
  //      any more: the certificate re-scores what is left) ----
  __syncthreads();
  for (int i = tid; i < R; i += MN_FIN2_THREADS)
    L.key[i] = (lu[i] == DEAD) ? MN_EMPTY : (((u64)lu[i] << 32) | (u64)lv[i]);
  if (tid == 0) {
    cnt->finisher_steps = (int)(steps > 0x7FFFFFFF ? 0x7FFFFFFF : steps);
    cnt->finisher_merges = merges;
    cnt->n_merged = merges;
    cnt->merged_S = merged_S;
    cnt->merged_E = merged_E;
  }
}

// `spec` set (components mode): the host has not seen the record count yet.  R is read from
// spec[0]; if it exceeds spec_limit, or the separability check counted violations (spec[1][0] != 0),
// the kernel does nothing and the host, which learns both at its one synchronisation, redoes the
// image on the ordinary path.
__global__ __launch_bounds__(MN_FIN2_THREADS) void mn_finisher_lds(
    ImgParams P, ObjState S, RecList L, int R, int* __restrict__ maprec, int* __restrict__ lists,
    Counters* __restrict__ cnt, long long max_steps, const int* __restrict__ spec_records,
    const int* __restrict__ spec_violations, int spec_limit, int* __restrict__ lcount) {
  if (spec_records) {
    R = *spec_records;
    if (R > spec_limit || spec_violations[0] != 0 || spec_violations[1] != 0) return;   // uniform
  }
  mn_fin_lds_run(P, S, L, R, maprec, lists, cnt, max_steps, lcount);
}
