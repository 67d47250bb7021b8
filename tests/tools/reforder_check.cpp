// Host build of mergenet_amd/csrc/mn_reforder.h (TEST HARNESS: the product runs the same text on the GPU).
//   g++ -O2 -std=c++17 -shared -fPIC -ffp-contract=off tests/tools/reforder_check.cpp -o tests/tools/libreforder_host.so
// Exports:
//   reforder_containers_check(seed, ops, every)  the flat-array hash map and heap against std::unordered_map and
//                                         std::priority_queue, operation by operation: iteration order of the map
//                                         after every `every`-th step, top of the heap after every push / pop.  0 = equal.
//   reforder_host_run(...)                the whole merge of one image with the reference's arithmetic (glibc logf,
//                                         double log for log(1 - p)); writes the partition (survivor per pixel).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <queue>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../mergenet_amd/csrc/mn_reforder.h"

namespace {

struct Owned {
  RoState S;
  std::vector<int> osize, ocls, parent, bcount, nelem, head, single, barena, nnext, r1, r2;
  std::vector<float> lp, oml, prio;
  std::vector<long long> boff, ctl;
  std::vector<unsigned long long> nkey, heap;
  void alloc(int N, int C, long long NL, long long arena, long long hcap) {
    osize.assign(N, 1); ocls.assign(N, 0); parent.resize(N); bcount.assign(N, 1); nelem.assign(N, 0);
    head.assign(N, MN_RO_NULL); single.assign(N, MN_RO_NULL); boff.assign(N, 0); barena.assign(arena, MN_RO_NULL);
    nnext.assign(2 * NL, MN_RO_NULL); nkey.assign(2 * NL, 0);
    r1.assign(NL, -1); r2.assign(NL, -1); oml.assign(NL, 0.0f); prio.assign(NL, -1.0f);
    heap.assign(hcap, 0ull); lp.assign((size_t)N * C, 0.0f); ctl.assign(16, 0);
    for (int i = 0; i < N; i++) parent[i] = i;
    S.N = N; S.C = C; S.NL = NL; S.omf = 1.0f; S.bias = 0.0f;
    S.osize = osize.data(); S.ocls = ocls.data(); S.lp = lp.data(); S.parent = parent.data();
    S.bcount = bcount.data(); S.nelem = nelem.data(); S.head = head.data(); S.boff = boff.data();
    S.single = single.data(); S.barena = barena.data(); S.barena_cap = arena;
    S.nnext = nnext.data(); S.nkey = nkey.data();
    S.r1 = r1.data(); S.r2 = r2.data(); S.oml = oml.data(); S.prio = prio.data();
    S.heap = heap.data(); S.hcap = hcap; S.ctl = ctl.data();
  }
};

struct ByPriority {
  bool operator()(const std::pair<float, int>& a, const std::pair<float, int>& b) const { return a.first < b.first; }
};

uint64_t rng_state;
uint64_t rng() {
  rng_state += 0x9E3779B97F4A7C15ull;
  uint64_t z = rng_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace

extern "C" int reforder_containers_check(unsigned long long seed, int ops, int every) {
  if (every < 1) every = 1;
  rng_state = seed;
  // --- the hash map: ONE object's map, keys of the form the reference uses, random inserts / erases ---
  {
    const int NODES = ops + 16;
    Owned W;
    W.alloc(1, 1, NODES, 64LL * NODES + 1024, 16);
    std::unordered_map<size_t, int> ref;
    std::vector<unsigned long long> present;
    int next_node = 0;
    for (int step = 0; step < ops; step++) {
      const bool ins = present.empty() || (rng() % 100) < (step < ops / 2 ? 70u : 40u);
      if (ins) {
        unsigned long long key;
        do {
          const int a = (int)(rng() % 4096), b = a + 1 + (int)(rng() % 4096);
          key = mn_ro_key(a, b);
        } while (ref.count((size_t)key));
        const int node = next_node++;
        if (!mn_ro_insert(W.S, 0, node, key)) return -1;
        ref[(size_t)key] = node;
        present.push_back(key);
      } else {
        const size_t i = (size_t)(rng() % present.size());
        const unsigned long long key = present[i];
        present[i] = present.back(); present.pop_back();
        const int n = mn_ro_erase(W.S, 0, key);
        if (n == MN_RO_NULL || n != ref[(size_t)key]) return 1000 + step;
        ref.erase((size_t)key);
      }
      if ((long long)ref.bucket_count() != (long long)W.S.bcount[0]) return 2000000 + step;
      if (step % every != 0 && step != ops - 1) continue;          // (the full walk is O(size))
      int n = W.S.head[0];
      for (std::unordered_map<size_t, int>::iterator it = ref.begin(); it != ref.end(); ++it) {
        if (n == MN_RO_NULL || n != it->second || W.S.nkey[n] != (unsigned long long)it->first) return 3000000 + step;
        n = W.S.nnext[n];
      }
      if (n != MN_RO_NULL) return 4000000 + step;
      // find: a present key and an absent one
      if (!present.empty()) {
        int prev;
        const unsigned long long key = present[(size_t)(rng() % present.size())];
        if (mn_ro_find(W.S, 0, key, &prev) != ref[(size_t)key]) return 5000000 + step;
        if (mn_ro_find(W.S, 0, key + 1, &prev) != MN_RO_NULL && !ref.count((size_t)(key + 1))) return 6000000 + step;
      }
    }
  }
  // --- the heap: few distinct priorities, so that ties are everywhere ---
  {
    Owned W;
    W.alloc(1, 1, 1, 64, ops + 16);
    std::priority_queue<std::pair<float, int>, std::vector<std::pair<float, int> >, ByPriority> ref;
    for (int step = 0; step < ops; step++) {
      const bool push = ref.empty() || (rng() % 100) < (step < ops / 2 ? 65u : 40u);
      if (push) {
        const float pr = (float)(rng() % 7) * 0.125f;
        if (!mn_ro_push(W.S, pr, step)) return -2;
        ref.push(std::make_pair(pr, step));
      } else {
        float pr; int rec;
        mn_ro_pop(W.S, &pr, &rec);
        if (pr != ref.top().first || rec != ref.top().second) return 7000000 + step;
        ref.pop();
      }
      if ((long long)ref.size() != W.S.ctl[1]) return 8000000 + step;
      if (!ref.empty() && (mn_ro_entry_prio(W.S.heap[0]) != ref.top().first || mn_ro_entry_rec(W.S.heap[0]) != ref.top().second)) return 9000000 + step;
    }
  }
  return 0;
}

// One image.  class_pred [C][H][W], adj_pred [O][H][W] (already clipped), offsets (drow, dcol); partition out.
// stats: [0] pops, [1] merges, [2] status, [3] bucket arena entries used, [4] largest queue size.
extern "C" int reforder_host_run(const float* class_pred, int C, const float* adj_pred, int O, int Wd, int H,
                                 const int* offs, float sdb, float omf, float bias, int* partition,
                                 int* object_class_of_root, long long* stats) {
  const int N = Wd * H;
  const long long NL = (long long)N * O;
  Owned W;
  W.alloc(N, C, NL, (long long)N * 256 + 65536, 8 * NL + 65536);
  RoState& S = W.S;
  S.omf = omf; S.bias = bias;
  for (int p = 0; p < N; p++) {
    int best = 0;
    for (int c = 0; c < C; c++) {
      const float l = logf(class_pred[(size_t)c * N + p]);
      S.lp[(size_t)p * C + c] = l;
      if (l > S.lp[(size_t)p * C + best]) best = c;
    }
    S.ocls[p] = best;
  }
  for (int p = 0; p < N; p++) {
    const int row = p / Wd, col = p % Wd;
    for (int k = 0; k < O; k++) {
      const int rr = row + offs[2 * k], cc = col + offs[2 * k + 1];
      if (rr < 0 || rr >= H || cc < 0 || cc >= Wd) continue;
      const int q = rr * Wd + cc;
      const long long r = (long long)p * O + k;
      float sp = adj_pred[(size_t)k * N + p];
      if (sdb != 0.0f) {
        const float logit = logf(sp) - log(1.0 - sp) + sdb;
        sp = 1.0 / (1.0 + expf(-logit));
      }
      S.r1[r] = p < q ? p : q; S.r2[r] = p < q ? q : p;
      S.oml[r] = logf(sp) - (float)log(1.0 - sp);
      int mc;
      S.prio[r] = mn_ro_score(S, (int)r, &mc);
    }
  }
  int rc = mn_ro_init(S, O, 1LL << 62);
  if (rc == MN_RO_DONE) rc = mn_ro_run(S, 1LL << 62);
  for (int p = 0; p < N; p++) {
    int x = p;
    while (S.parent[x] != x) x = S.parent[x];
    partition[p] = x;
    if (object_class_of_root) object_class_of_root[p] = S.ocls[x];
  }
  if (stats) { stats[0] = S.ctl[3]; stats[1] = S.ctl[4]; stats[2] = rc; stats[3] = S.ctl[2]; stats[4] = S.ctl[6]; }
  return rc == MN_RO_DONE ? 0 : rc;
}
