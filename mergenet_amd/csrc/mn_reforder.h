// mn_reforder.h -- the reference's merge loop WITH its order among bit-equal priorities.
//
// MN_MODE_EXACT pops "the largest stored priority, lowest record id".  The reference pops whatever its
// std::priority_queue holds on top (segment.h:270-275 compares the float only, so among equals the position in
// the binary heap decides) and folds the absorbed object's records in the iteration order of an
// std::unordered_map keyed id1 * 1619 + id2 * 3203 (segment.cc:650-652, segment.h:237-242), which is the order
// of the pushes.  On maps with plateaus of equal values (clipped, blurred with a wide kernel) that order
// decides instance borders.  This file restates, on flat arrays, exactly the two
// containers as the libstdc++ of this toolchain (GCC 11) behaves -- it is that build of the reference the golden
// vectors come from:
//   * binary heap: std::push_heap / std::pop_heap (bits/stl_heap.h: __push_heap, __adjust_heap), entries
//     (priority, record), stale entries left in place as the reference leaves them (segment.cc:553-558);
//   * hash map with unique keys, identity hash, hash codes not cached: one singly linked list of all nodes,
//     buckets[] pointing at the node BEFORE a bucket's first node (bits/hashtable.h: _M_insert_bucket_begin,
//     _M_erase / _M_remove_bucket_begin, _M_rehash_aux), growth by _Prime_rehash_policy: 1 -> 13 -> 29 -> 59 ...
//     (the chain below was read off the library: tests/tools/reforder_check.cpp checks this model against
//     std::unordered_map itself, operation by operation).
// Everything is sequential (ONE lane; a step is a chain of dependent memory accesses), an order of magnitude
// slower per step than the exact engine: meant for small images and for callers who need the reference's
// very partition on tie-decided inputs.  Host and device compile the same text (MN_REF_HD).
#pragma once
#include <stdint.h>

#ifndef MN_REF_HD
#if defined(__HIPCC__)
#define MN_REF_HD __host__ __device__ __forceinline__
#else
#define MN_REF_HD static inline
#endif
#endif

#define MN_RO_NULL (-1)            /* bucket: no node of this bucket yet;  next: end of the list */
#define MN_RO_BEFORE_BEGIN (-2)    /* bucket: its first node is the first node of the whole list */

enum { MN_RO_RUNNING = 0, MN_RO_DONE = 1, MN_RO_BUDGET = 2, MN_RO_ARENA_FULL = 3, MN_RO_HEAP_FULL = 4,
       MN_RO_CORRUPT = 5 };

struct RoState {
  int N, C;
  long long NL;                    // record slots (pixel * O + k); dead slots have r1 < 0
  float omf, bias;
  // objects (Object, segment.h:85-137)
  int* osize; int* ocls; float* lp; int* parent;
  // one hash map per object (Object::adjacency_list)
  int* bcount;                     // bucket count (1 = the single-bucket state of an empty map)
  int* nelem;
  int* head;                       // first node of the list (before_begin.next) or MN_RO_NULL
  long long* boff;                 // this object's bucket array in barena (bcount > 1)
  int* single;                     // the one bucket of the initial state
  int* barena; long long barena_cap;
  // nodes: two per record (one in the map of each end)
  int* nnext; unsigned long long* nkey;
  // records (AdjacencyRecord, segment.h:175-232)
  int* r1; int* r2;                // ends, r1 < r2 by id; r2 = -1: merged away (obj2 = NULL, segment.cc:726)
  float* oml; float* prio;
  // the queue (segmenter_queue)
  unsigned long long* heap; long long hcap;     // entry = (record << 32) | bits of the float priority: ONE access
  // progress (kept in memory so that a launch can stop and the next one go on)
  long long* ctl;                  // [0] status, [1] heap size, [2] arena bump, [3] pops, [4] merges, [5] init cursor, [6] largest heap size
};

// bucket counts a map grows through when elements arrive one at a time (_Prime_rehash_policy::_M_next_bkt(2n))
MN_REF_HD long long mn_ro_next_bcount(long long bc) {
  const long long chain[] = {1, 13, 29, 59, 127, 257, 541, 1109, 2357, 5087, 10273, 20753, 42043, 85229, 172933, 351061,
                             712697, 1447153, 2938679, 5967347, 12117689, 24607243, 49969847, 101473717, 206062531,
                             418451333, 849749479, 1725587117};
  for (int i = 0; i + 1 < (int)(sizeof(chain) / sizeof(chain[0])); i++)
    if (chain[i] == bc) return chain[i + 1];
  return -1;
}

MN_REF_HD unsigned long long mn_ro_entry(float pr, int rec) {
  union { float f; unsigned u; } c;
  c.f = pr;
  return ((unsigned long long)(unsigned)rec << 32) | (unsigned long long)c.u;
}
MN_REF_HD float mn_ro_entry_prio(unsigned long long e) {
  union { float f; unsigned u; } c;
  c.u = (unsigned)(e & 0xFFFFFFFFull);
  return c.f;
}
MN_REF_HD int mn_ro_entry_rec(unsigned long long e) { return (int)(e >> 32); }

MN_REF_HD int* mn_ro_buckets(const RoState& S, int o) {
  return S.bcount[o] == 1 ? &S.single[o] : S.barena + S.boff[o];
}

MN_REF_HD unsigned long long mn_ro_key(int a, int b) {        // AdjacencyRecordHasher, a < b
  return (unsigned long long)a * 1619ull + (unsigned long long)b * 3203ull;
}

// _M_find_before_node: the node holding `key` in o's map (or MN_RO_NULL) and the node before it
MN_REF_HD int mn_ro_find(const RoState& S, int o, unsigned long long key, int* prev_out) {
  const unsigned long long bc = (unsigned long long)S.bcount[o];
  const int b = (int)(key % bc);
  const int* bk = mn_ro_buckets(S, o);
  int prev = bk[b];
  if (prev == MN_RO_NULL) return MN_RO_NULL;
  int n = prev == MN_RO_BEFORE_BEGIN ? S.head[o] : S.nnext[prev];
  for (;;) {
    if (S.nkey[n] == key) { *prev_out = prev; return n; }
    const int nx = S.nnext[n];
    if (nx == MN_RO_NULL || (int)(S.nkey[nx] % bc) != b) return MN_RO_NULL;
    prev = n;
    n = nx;
  }
}

// _M_rehash_aux (unique keys): every node, in list order, goes to the front of its new bucket; a bucket
// seen for the first time goes to the front of the whole list
MN_REF_HD bool mn_ro_rehash(RoState& S, int o, long long nbc) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (several lanes may grow different objects' maps at once: where an array lands does not matter)
  const long long bump = (long long)atomicAdd(reinterpret_cast<unsigned long long*>(&S.ctl[2]), (unsigned long long)nbc);
  if (bump + nbc > S.barena_cap) return false;
#else
  const long long bump = S.ctl[2];
  if (bump + nbc > S.barena_cap) return false;
  S.ctl[2] = bump + nbc;
#endif
  int* nb = S.barena + bump;
  for (long long i = 0; i < nbc; i++) nb[i] = MN_RO_NULL;
  int p = S.head[o];
  int first = MN_RO_NULL;                               // before_begin.next
  long long bbegin_bkt = 0;
  while (p != MN_RO_NULL) {
    const int next = S.nnext[p];
    const long long b = (long long)(S.nkey[p] % (unsigned long long)nbc);
    if (nb[b] == MN_RO_NULL) {
      S.nnext[p] = first;
      first = p;
      nb[b] = MN_RO_BEFORE_BEGIN;
      if (S.nnext[p] != MN_RO_NULL) nb[bbegin_bkt] = p;
      bbegin_bkt = b;
    } else {
      const int before = nb[b];
      if (before == MN_RO_BEFORE_BEGIN) { S.nnext[p] = first; first = p; }
      else { S.nnext[p] = S.nnext[before]; S.nnext[before] = p; }
    }
    p = next;
  }
  S.head[o] = first;
  S.boff[o] = bump;
  S.bcount[o] = (int)nbc;
  return true;
}

// operator[] with a key that is not in the map: _M_insert_unique_node (rehash check first, then the node
// goes to the beginning of its bucket)
MN_REF_HD bool mn_ro_insert(RoState& S, int o, int node, unsigned long long key) {
  const long long bc = S.bcount[o], ne = S.nelem[o];
  // _M_need_rehash(bc, ne, 1): _M_next_resize is 0 for the untouched map, bc afterwards (max_load_factor 1)
  if (ne + 1 > (bc == 1 ? 0 : bc)) {
    const long long nbc = mn_ro_next_bcount(bc);
    if (nbc < 0 || !mn_ro_rehash(S, o, nbc)) return false;
  }
  S.nkey[node] = key;
  const unsigned long long ubc = (unsigned long long)S.bcount[o];
  int* bk = mn_ro_buckets(S, o);
  const int b = (int)(key % ubc);
  if (bk[b] != MN_RO_NULL) {
    const int before = bk[b];
    if (before == MN_RO_BEFORE_BEGIN) { S.nnext[node] = S.head[o]; S.head[o] = node; }
    else { S.nnext[node] = S.nnext[before]; S.nnext[before] = node; }
  } else {
    S.nnext[node] = S.head[o];
    S.head[o] = node;
    if (S.nnext[node] != MN_RO_NULL) bk[(int)(S.nkey[S.nnext[node]] % ubc)] = node;
    bk[b] = MN_RO_BEFORE_BEGIN;
  }
  S.nelem[o] = (int)ne + 1;
  return true;
}

// erase(key): _M_erase(bkt, prev, n) with _M_remove_bucket_begin.  Returns the node (now free) or MN_RO_NULL.
MN_REF_HD int mn_ro_erase(RoState& S, int o, unsigned long long key) {
  int prev = MN_RO_NULL;
  const int n = mn_ro_find(S, o, key, &prev);
  if (n == MN_RO_NULL) return MN_RO_NULL;
  const unsigned long long bc = (unsigned long long)S.bcount[o];
  int* bk = mn_ro_buckets(S, o);
  const int b = (int)(key % bc);
  const int next = S.nnext[n];
  const int next_bkt = next != MN_RO_NULL ? (int)(S.nkey[next] % bc) : 0;
  if (prev == bk[b]) {
    // n is the first node of its bucket
    if (next == MN_RO_NULL || next_bkt != b) {
      if (next != MN_RO_NULL) bk[next_bkt] = bk[b];
      // (if the bucket began the list, before_begin.next = next: done by the unlink below)
      bk[b] = MN_RO_NULL;
    }
  } else if (next != MN_RO_NULL && next_bkt != b) {
    bk[next_bkt] = prev;
  }
  if (prev == MN_RO_BEFORE_BEGIN) S.head[o] = next; else S.nnext[prev] = next;
  S.nelem[o]--;
  return n;
}

// std::priority_queue::push: push_back + __push_heap
MN_REF_HD bool mn_ro_push(RoState& S, float pr, int rec) {
  long long hole = S.ctl[1];
  if (hole >= S.hcap) return false;
  S.ctl[1] = hole + 1;
  if (hole + 1 > S.ctl[6]) S.ctl[6] = hole + 1;        // (largest queue so far)
  while (hole > 0) {
    const long long par = (hole - 1) / 2;
    const unsigned long long e = S.heap[par];
    if (!(mn_ro_entry_prio(e) < pr)) break;
    S.heap[hole] = e;
    hole = par;
  }
  S.heap[hole] = mn_ro_entry(pr, rec);
  return true;
}

// top() + pop(): __pop_heap moves the last entry's value down from the root with __adjust_heap (to the bottom
// along the larger children -- the RIGHT one among equals -- then back up with __push_heap)
MN_REF_HD void mn_ro_pop(RoState& S, float* pr_out, int* rec_out) {
  const long long n = S.ctl[1];
  *pr_out = mn_ro_entry_prio(S.heap[0]); *rec_out = mn_ro_entry_rec(S.heap[0]);
  const long long len = n - 1;
  S.ctl[1] = len;
  if (len == 0) return;
  const unsigned long long v = S.heap[len];
  const float vp = mn_ro_entry_prio(v);
  long long hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (mn_ro_entry_prio(S.heap[child]) < mn_ro_entry_prio(S.heap[child - 1])) child--;
    S.heap[hole] = S.heap[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    S.heap[hole] = S.heap[child - 1];
    hole = child - 1;
  }
  while (hole > 0) {
    const long long par = (hole - 1) / 2;
    const unsigned long long e = S.heap[par];
    if (!(mn_ro_entry_prio(e) < vp)) break;
    S.heap[hole] = e;
    hole = par;
  }
  S.heap[hole] = v;
}

// ComputeClassDeltaLogprob + UpdateMergePriority (segment.cc:107-150), the reference's float32 operation order
MN_REF_HD float mn_ro_score(const RoState& S, int r, int* mcls_out) {
  const int a = S.r1[r], b = S.r2[r];
  float cdl = 0.0f;
  int mc = S.ocls[a];
  if (S.ocls[a] != S.ocls[b]) {
    const float* la = S.lp + (size_t)a * S.C;
    const float* lb = S.lp + (size_t)b * S.C;
    int best = 0;
    float bestv = la[0] + lb[0];
    for (int c = 1; c < S.C; c++) {
      const float v = la[c] + lb[c];
      if (v > bestv) { bestv = v; best = c; }
    }
    mc = best;
    cdl = (bestv - la[S.ocls[a]]) - lb[S.ocls[b]];
  }
  *mcls_out = mc;
  const float den = (float)((unsigned long long)S.osize[a] + (unsigned long long)S.osize[b]);
  return (S.oml[r] * S.omf + cdl) / den + S.bias;
}

// The constructor's loop (segment.cc:209-231) from `cursor` on: records in creation order into the maps of
// both ends (the source pixel's first) and, if >= 0, into the queue.  `src[r]` is the source pixel of slot r.
MN_REF_HD int mn_ro_init_record(RoState& S, int O, long long r) {        // record slot r into the maps of both ends
  const int p = (int)(r / O);                            // the source pixel of slot r = pixel * O + k
  const int a = S.r1[r], b = S.r2[r];
  const int q = a == p ? b : a;
  const unsigned long long key = mn_ro_key(a, b);
  if (!mn_ro_insert(S, p, (int)(2 * r), key)) return MN_RO_ARENA_FULL;
  if (!mn_ro_insert(S, q, (int)(2 * r + 1), key)) return MN_RO_ARENA_FULL;
  return MN_RO_RUNNING;
}

MN_REF_HD int mn_ro_init(RoState& S, int O, long long budget) {
  long long r = S.ctl[5];
  for (; r < S.NL && budget > 0; r++, budget--) {
    if (S.r1[r] < 0) continue;
    const int rc = mn_ro_init_record(S, O, r);
    if (rc != MN_RO_RUNNING) { S.ctl[5] = r; return rc; }
    if (S.prio[r] >= 0.0f && !mn_ro_push(S, S.prio[r], (int)r)) { S.ctl[5] = r; return MN_RO_HEAP_FULL; }
  }
  S.ctl[5] = r;
  return r >= S.NL ? MN_RO_DONE : MN_RO_BUDGET;
}

// Merge (segment.cc:602-727) in three pieces, so that the device can put a wave-wide push between the records
// of the walk: begin (survivor's state, the merged record leaves both maps; returns the first node of the
// absorbed object's list), one record of the walk (may ask for ONE push: *push_rec >= 0), end.
MN_REF_HD int mn_ro_merge_begin(RoState& S, int r, int mcls, int* a_out, int* b_out, int* first) {
  int a = S.r1[r], b = S.r2[r];
  if (S.osize[a] < S.osize[b]) { const int t = a; a = b; b = t; }     // the larger survives, a tie keeps r1
  S.ocls[a] = mcls;
  S.osize[a] += S.osize[b];
  float* la = S.lp + (size_t)a * S.C;
  const float* lb = S.lp + (size_t)b * S.C;
  for (int c = 0; c < S.C; c++) la[c] += lb[c];
  const unsigned long long rk = mn_ro_key(S.r1[r], S.r2[r]);
  if (mn_ro_erase(S, a, rk) == MN_RO_NULL) return MN_RO_CORRUPT;
  if (mn_ro_erase(S, b, rk) == MN_RO_NULL) return MN_RO_CORRUPT;
  *a_out = a; *b_out = b; *first = S.head[b];
  return MN_RO_RUNNING;
}

// `defer_a` (device: the records of a walk go to one lane each): the insert into the SURVIVOR's map -- the one
// step whose order among the records of a walk matters -- is left to the caller (*adopt_key != 0).  Everything
// else touches the record itself and the third object's map, which no other record of this walk touches.
MN_REF_HD int mn_ro_merge_node(RoState& S, int a, int b, int it, int* next, float* push_prio, int* push_rec,
                               int defer_a = 0, unsigned long long* adopt_key = 0) {
  *next = S.nnext[it];                                   // (the iterator's increment: this node's slot is reused below)
  *push_rec = -1;
  if (adopt_key) *adopt_key = 0ull;
  const int t = it >> 1;
  int c3;
  const unsigned long long old_key = mn_ro_key(S.r1[t], S.r2[t]);
  if (S.r1[t] == b) c3 = S.r2[t]; else c3 = S.r1[t];
  const int lo = a < c3 ? a : c3, hi = a < c3 ? c3 : a;
  const unsigned long long new_key = mn_ro_key(lo, hi);
  const int n3 = mn_ro_erase(S, c3, old_key);
  if (n3 == MN_RO_NULL) return MN_RO_CORRUPT;
  int prev;
  const int hit = mn_ro_find(S, a, new_key, &prev);
  int mc;
  if (hit != MN_RO_NULL) {
    const int u = hit >> 1;
    S.oml[u] += S.oml[t];
    S.prio[t] = 1.17549435e-38f;                         // numeric_limits<float>::min(): never equals a queue entry
    S.r1[t] = lo; S.r2[t] = hi;                          // (the folded record keeps two live ends, segment.cc:658-663)
    S.prio[u] = mn_ro_score(S, u, &mc);
    if (S.prio[u] >= 0.0f) { *push_prio = S.prio[u]; *push_rec = u; }
  } else {
    S.r1[t] = lo; S.r2[t] = hi;
    if (defer_a) *adopt_key = new_key;                                     // (keys are never 0: hi >= 1)
    else if (!mn_ro_insert(S, a, it, new_key)) return MN_RO_ARENA_FULL;    // the slot b's map held
    if (!mn_ro_insert(S, c3, n3, new_key)) return MN_RO_ARENA_FULL;
    S.prio[t] = mn_ro_score(S, t, &mc);
    if (S.prio[t] >= 0.0f) { *push_prio = S.prio[t]; *push_rec = t; }
  }
  return MN_RO_RUNNING;
}

MN_REF_HD void mn_ro_merge_end(RoState& S, int r, int a, int b) {
  S.head[b] = MN_RO_NULL; S.nelem[b] = 0;
  S.parent[b] = a;
  S.r2[r] = -1;
  S.ctl[4]++;
}

MN_REF_HD int mn_ro_merge(RoState& S, int r, int mcls) {
  int a, b, it;
  int rc = mn_ro_merge_begin(S, r, mcls, &a, &b, &it);
  if (rc != MN_RO_RUNNING) return rc;
  while (it != MN_RO_NULL) {
    int nx, prec; float pp;
    rc = mn_ro_merge_node(S, a, b, it, &nx, &pp, &prec);
    if (rc != MN_RO_RUNNING) return rc;
    if (prec >= 0 && !mn_ro_push(S, pp, prec)) return MN_RO_HEAP_FULL;
    it = nx;
  }
  mn_ro_merge_end(S, r, a, b);
  return MN_RO_RUNNING;
}

// RunSegmentation (segment.cc:539-573), at most `budget` pops
MN_REF_HD int mn_ro_run(RoState& S, long long budget) {
  while (S.ctl[1] > 0) {
    if (budget-- <= 0) return MN_RO_BUDGET;
    float q; int r;
    mn_ro_pop(S, &q, &r);
    S.ctl[3]++;
    if (q != S.prio[r]) continue;
    if (S.r2[r] < 0) continue;
    int mc;
    const float f = mn_ro_score(S, r, &mc);
    S.prio[r] = f;
    if (f == q) {
      const int rc = mn_ro_merge(S, r, mc);
      if (rc != MN_RO_RUNNING) return rc;
    } else if (f >= 0.0f) {
      if (!mn_ro_push(S, f, r)) return MN_RO_HEAP_FULL;
    }
  }
  return MN_RO_DONE;
}
