// exact_model.cpp -- TEST / DESIGN TOOL, NOT PRODUCT CODE.
//
// CPU model of the exact engine's SEMANTICS (mergenet_amd/csrc/mn_kernels_exact.h): the reference's lazy
// greedy (utils/csegment/segment.cc:539-573, 602-727) with its float32 arithmetic, but with the queue as
// "the stored priority of every live record, if >= 0" under a TOTAL order: priority, then record id
// (= pixel * O + offset index, the creation order of segment.cc:209-231).  Uses the C library's logf /
// log, so its numbers are the reference's; differs from oracle/csegment_oracle.cpp only in the order
// among bit-equal priorities (the oracle keeps std::priority_queue's heap order).
//
// Used to (a) find inputs whose result the tie rule decides, without a GPU (tests/golden/make_golden.py
// --tie-search), (b) measure the structure of the event sequence (nesting depth, super-events, tie
// conflicts) that DESIGN.md section 4.2 / 5 quotes, (c) check the tie-conflict criterion of the engine
// (stats.tied_conflicts) against an independent implementation.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC exact_model.cpp -o libexact_model.so
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <set>
#include <unordered_map>
#include <vector>

namespace {

struct Key {
  float p; int r;
  bool operator<(const Key& o) const { return p > o.p || (p == o.p && r < o.r); }   // largest priority, lowest id first
};

struct Model {
  int C, O, W, H, N;
  float omf, bias;
  std::vector<float> lp;
  std::vector<int> ocls, osize, parent;
  std::vector<std::unordered_map<int, int>> adj;      // other endpoint -> record
  std::vector<int> r1, r2;                            // endpoints (r1 < r2), r2 = -1: dead
  std::vector<float> S, prio;                         // log-odds sum, stored priority (NaN-free; < 0: not queued)
  std::set<Key> q;
  long long steps = 0, merges = 0, tied_steps = 0, tied_merges = 0;
  long long repop_merges = 0, refresh_then_other_refresh = 0, refresh_then_other_merge = 0; int last_refreshed = -1;
  // ---- tie-conflict criterion ----
  // Events form a forest by nesting: event j hangs under the last earlier event i with word(i) <= word(j)
  // such that everything between them is above word(i) (the suffix minima of the popped priorities).
  // Two events with EQUAL priorities that are siblings in that forest are ordered by the tie rule; their
  // subtrees commute iff they touch disjoint objects.  stamp[o] = index of the last event that touched o.
  std::vector<long long> stamp;
  std::vector<std::pair<float, long long>> stk;        // suffix minima: (priority, event index), priorities non-decreasing
  long long tied_pairs_on_stack = 0;                   // adjacent stack entries with equal priority
  long long tied_conflicts = 0;
  long long max_depth = 0;
  int track = 0;
  // ---- how much could run side by side?  (design aid: XM_PARALLEL=1)  A ROOT is a pop at a new minimum of the popped
  // priorities (stack depth 1); its super-event is everything popped until the next root.  Consecutive super-events
  // with pairwise disjoint footprints (objects touched) commute: they could run on different waves. ----
  int parallel_study = 0;
  long long n_roots = 0, cur_root_events = 0, max_root_events = 0;
  std::vector<int> cur_fp;                                 // objects touched by the current super-event
  std::vector<long long> batch_mark;                       // per object: id of the batch that last claimed it
  long long batch_id = 1, batch_len = 0, n_batches = 0, sum_batch_events = 0, cur_batch_events = 0;
  std::vector<long long> batch_hist = std::vector<long long>(16, 0);   // batch length (super-events), log2 buckets
  std::vector<std::vector<int>> all_fp;                    // every super-event's footprint, in order (window study)
  void close_super_event() {
    if (cur_root_events == 0) return;
    std::sort(cur_fp.begin(), cur_fp.end());
    cur_fp.erase(std::unique(cur_fp.begin(), cur_fp.end()), cur_fp.end());
    bool clash = false;
    for (int o : cur_fp) if (batch_mark[o] == batch_id) { clash = true; break; }
    if (clash) {
      int b = 0; while ((1LL << (b + 1)) <= batch_len && b < 15) b++;
      batch_hist[b]++; n_batches++;
      batch_id++; batch_len = 0;
    }
    for (int o : cur_fp) batch_mark[o] = batch_id;
    batch_len++;
    if (parallel_study >= 2) all_fp.push_back(cur_fp);
    if (cur_root_events > max_root_events) max_root_events = cur_root_events;
    cur_fp.clear(); cur_root_events = 0;
  }
  // ---- a PREFIX executor (design aid, XM_ROUNDS=K): each round looks at the K best queue entries as they stand and
  // commits the longest prefix of the SEQUENTIAL pop sequence that (a) pops exactly those entries, in their order,
  // with the priorities they had at the start of the round (no record that an earlier pop of the round stored or
  // re-scored comes in between), and (b) has pairwise compatible footprints (no object written by one and read or
  // written by another).  Such a prefix can be executed side by side with the sequential result, bit for bit. ----
  int rounds_k = 0;
  std::vector<Key> rs_snap, rs_pre;
  size_t rs_pos = 0;
  std::vector<long long> rs_wmark, rs_rmark;
  std::vector<int> rs_w, rs_r;
  long long rs_round = 1, rs_rounds = 0, rs_in_round = 0, rs_cut_order = 0, rs_cut_clash = 0, rs_cut_window = 0;
  std::vector<long long> rs_hist = std::vector<long long>(12, 0);
  bool rs_unavailable = false;
  void rs_begin(const Key& top) {
    rs_unavailable = !(rs_pos < rs_snap.size() && rs_snap[rs_pos].r == top.r && rs_snap[rs_pos].p == top.p);
    rs_pre.clear();
    auto it = q.begin();
    for (int i = 0; i < rounds_k && it != q.end(); ++i, ++it) rs_pre.push_back(*it);
    rs_w.clear(); rs_r.clear();
  }
  void rs_end() {
    bool clash = false;
    for (int o : rs_w) if (rs_wmark[o] == rs_round || rs_rmark[o] == rs_round) { clash = true; break; }
    if (!clash) for (int o : rs_r) if (rs_wmark[o] == rs_round) { clash = true; break; }
    if (rs_unavailable || clash) {
      if (rs_in_round > 0) {
        rs_rounds++;
        int b = 0; while ((1LL << (b + 1)) <= rs_in_round && b < 11) b++;
        rs_hist[b]++;
        if (rs_unavailable) { if (rs_pos >= rs_snap.size()) rs_cut_window++; else rs_cut_order++; } else rs_cut_clash++;
      }
      rs_round++; rs_in_round = 0;
      rs_snap = rs_pre; rs_pos = 0;
    }
    rs_pos++; rs_in_round++;
    for (int o : rs_w) { rs_wmark[o] = rs_round; rs_rmark[o] = rs_round; }
    for (int o : rs_r) rs_rmark[o] = rs_round;
  }
  // ---- arena accounting (design aid): what the engine's adjacency arena would need under a policy ----
  // ---- pair-table accounting (design aid, XM_REKEY=1): live records whose key is no longer the pixel pair they
  // were created with -- what a pair table would hold if records with their original key were found by arithmetic ----
  int rekey_study = 0;
  std::vector<unsigned char> rekeyed;
  long long live_rekeyed = 0, peak_rekeyed = 0, live_at_peak = 0, live_records = 0, n_start = 0;
  int cap0 = 64, growth = 4, round_to = 64, reuse = 0, growth_pct = 0;
  std::vector<int> acap, alen;                          // per object: capacity, entries in use (dead ones included)
  long long bump = 0, reallocs = 0, moved = 0;
  std::vector<std::vector<long long>> freelist;         // by size class (log2)
  int cls_of(int cap) const {
    if (reuse == 2) return cap / round_to;              // exact sizes (multiples of round_to)
    int c = 0; while ((1 << c) < cap) c++; return c;
  }
  long long alloc_block(int cap) {
    if (reuse) {
      const int c = cls_of(cap);
      if ((int)freelist.size() > c && !freelist[c].empty()) { long long p = freelist[c].back(); freelist[c].pop_back(); return p; }
    }
    const long long p = bump; bump += cap; return p;
  }
  void free_block(int cap) {
    if (!reuse || cap <= 0) return;
    const int c = cls_of(cap);
    if (reuse == 1 && (1 << c) != cap) return;          // (only power-of-two blocks are recycled)
    if ((int)freelist.size() <= c) freelist.resize(c + 1);
    freelist[c].push_back(0);
  }

  float score(int a, int b, float s, int* mc) const {
    float cdl = 0.0f;
    int m = ocls[a];
    if (ocls[a] != ocls[b]) {
      const float* la = &lp[(size_t)a * C];
      const float* lb = &lp[(size_t)b * C];
      int best = 0;
      float bestv = la[0] + lb[0];
      for (int c = 1; c < C; c++) { const float v = la[c] + lb[c]; if (v > bestv) { bestv = v; best = c; } }
      m = best;
      cdl = bestv - la[ocls[a]] - lb[ocls[b]];
    }
    *mc = m;
    const size_t den = (size_t)osize[a] + (size_t)osize[b];
    return (s * omf + cdl) / den + bias;
  }

  // ---- the sorted FRONT of the executor's queue (XM_FRONT=M with XM_EXECUTOR=K): the M best entries, with the
  // invariant "every entry outside is behind the front's last"; a store that beats the last goes in (the last
  // falls out when the front is full), a store of a record that is in the front takes it out, and the front is
  // rebuilt when fewer than K entries are left.  The model checks every round that the front's first K entries are
  // the queue's (fr_checks / fr_rebuilds / fr_inserts are reported). ----
  int front_m = 0;
  std::vector<Key> front;
  long long fr_rebuilds = 0, fr_inserts = 0, fr_checks = 0, fr_errors = 0;
  void front_remove(const Key& k) {
    auto it = std::lower_bound(front.begin(), front.end(), k);
    if (it != front.end() && it->r == k.r && it->p == k.p) front.erase(it);
  }
  void front_offer(const Key& k) {
    if (front.empty() || !(k < front.back())) return;            // behind the last: stays outside
    front.insert(std::lower_bound(front.begin(), front.end(), k), k);
    fr_inserts++;
    if ((int)front.size() > front_m) front.pop_back();
  }
  void front_rebuild() {
    front.clear();
    auto it = q.begin();
    for (int i = 0; i < front_m && it != q.end(); ++i, ++it) front.push_back(*it);
    fr_rebuilds++;
  }
  void store(int r, float f) {
    if (prio[r] >= 0.0f) { q.erase(Key{prio[r], r}); if (front_m) front_remove(Key{prio[r], r}); }
    prio[r] = f;
    if (f >= 0.0f) { q.insert(Key{f, r}); if (front_m) front_offer(Key{f, r}); }
  }

  // touch of object o by event `ev` (popped priority pw): conflict iff the last toucher lies in the subtree
  // of a tied sibling of one of ev's ancestors-or-self
  bool dangerous(long long s) const {
    if (s == 0 || tied_pairs_on_stack == 0) return false;
    // first stack entry with index > s
    size_t lo = 0, hi = stk.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (stk[mid].second > s) hi = mid; else lo = mid + 1; }
    if (lo == stk.size() || lo == 0) return false;     // (s is the current event itself, or older than the whole stack)
    return stk[lo - 1].first == stk[lo].first;
  }
  // rw = 0 (XM_RW=0, the first form of the criterion): every touch counts alike (an object touched from two tied subtrees is a conflict).
  // rw = 1: a touch either WRITES the object's state (the two ends of a merge) or only READS it (the ends of a pop that
  // stores a fresh priority; the third objects of a merge, whose records with the merged pair are rewritten): two
  // tied subtrees commute unless one writes an object the other reads or writes.  Every record an event rewrites has
  // an end the event writes and every record it reads has both ends touched, so records need no stamps of their own.
  // The read stamp is not overwritten while it is dangerous (a later read from the toucher's own subtree must not
  // hide it from a write that follows).
  int rw = 1;
  long long walk_hist[16] = {0}, walk_max = 0;        // records of the absorbed object at a merge, log2 buckets (XM_WALKS=1)
  long long kind_count[4] = {0, 0, 0, 0};              // conflicts: write after write, read after write, write after read, rival
  std::vector<long long> rstamp;
  void touch(int o, long long ev, bool write = true) {
    if (!track) return;
    if (!rw) {
      const long long s = stamp[o];
      stamp[o] = ev;
      if (dangerous(s)) tied_conflicts++;
      return;
    }
    if (dangerous(stamp[o])) { tied_conflicts++; kind_count[write ? 0 : 1]++; }
    if (write) {
      if (dangerous(rstamp[o])) { tied_conflicts++; kind_count[2]++; }
      stamp[o] = ev; rstamp[o] = ev;
    } else if (!dangerous(rstamp[o])) {
      rstamp[o] = ev;
    }
  }

  void sibling_check(float old) {
    if (!track || old < 0.0f) return;
    size_t lo = 0, hi = stk.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (stk[mid].first >= old) hi = mid; else lo = mid + 1; }
    if (lo < stk.size() && stk[lo].first == old) { tied_conflicts++; kind_count[3]++; }
  }

  // ---- the prefix executor itself (XM_EXECUTOR=K): NOT a count along the sequential order but the algorithm a
  // workgroup would run.  Per round: the K best queue entries; for each, from the state at the START of the round
  // and without changing anything, a PLAN (fresh priority; for a merge the survivor's new state and, per record of
  // the absorbed object, fold or adopt, new log-odds sum, new priority); then the longest prefix is committed such
  // that no committed plan produced a queue key ahead of the next entry and no plan writes an object that an earlier
  // committed plan read or wrote, or reads one it wrote.  The result must be the sequential model's, bit for bit
  // (tests/test_exact_model.py compares partitions, pops and merges). ----
  int executor_k = 0;
  long long ex_rounds = 0;
  struct PlanItem { int c3, t, tr; bool fold; float newS, newprio; };
  struct Plan {
    Key key; int r; bool merging; float f; int a, b, mc, newsize;
    std::vector<float> newlp;
    std::vector<PlanItem> items;
    bool has_best; Key best;
  };
  float score_as(const float* la, int cls_a, int size_a, int o, float s_, int* mc) const {
    float cdl = 0.0f;
    int m = cls_a;
    if (cls_a != ocls[o]) {
      const float* lb = &lp[(size_t)o * C];
      int best = 0;
      float bestv = la[0] + lb[0];
      for (int c = 1; c < C; c++) { const float v = la[c] + lb[c]; if (v > bestv) { bestv = v; best = c; } }
      m = best;
      cdl = bestv - la[cls_a] - lb[ocls[o]];
    }
    *mc = m;
    const size_t den = (size_t)size_a + (size_t)osize[o];
    return (s_ * omf + cdl) / den + bias;
  }
  void offer(Plan& P, float pr, int rid) const {
    if (!(pr >= 0.0f)) return;
    const Key k{pr, rid};
    if (!P.has_best || k < P.best) { P.best = k; P.has_best = true; }
  }
  void make_plan(const Key& e, Plan& P) const {
    P.key = e; P.r = e.r; P.items.clear(); P.newlp.clear(); P.has_best = false;
    const int x = r1[e.r], y = r2[e.r];
    P.f = score(x, y, S[e.r], &P.mc);
    P.merging = P.f == e.p;
    P.a = x; P.b = y;
    if (!P.merging) { offer(P, P.f, e.r); return; }
    if (osize[P.a] < osize[P.b]) std::swap(P.a, P.b);
    const int a = P.a, b = P.b;
    P.newsize = osize[a] + osize[b];
    P.newlp.resize(C);
    for (int c = 0; c < C; c++) P.newlp[c] = lp[(size_t)a * C + c] + lp[(size_t)b * C + c];
    for (const auto& kv : adj[b]) {
      const int c3 = kv.first, t = kv.second;
      if (c3 == a) continue;
      PlanItem it;
      it.c3 = c3; it.t = t;
      const auto hit = adj[a].find(c3);
      it.fold = hit != adj[a].end();
      it.tr = it.fold ? hit->second : t;
      it.newS = it.fold ? S[it.tr] + S[t] : S[t];
      int m2;
      // (the record's ends are ordered by id; the score is symmetric in them up to which class vector is `la`: the
      //  sequential code calls score(r1, r2) with r1 < r2 -- keep that order)
      if (a < c3) it.newprio = score_as(P.newlp.data(), P.mc, P.newsize, c3, it.newS, &m2);
      else it.newprio = score_rev(c3, P.newlp.data(), P.mc, P.newsize, it.newS, &m2);
      offer(P, it.newprio, it.tr);
      P.items.push_back(it);
    }
  }
  // score(o, A') with o < a: the first operand is the untouched object, the second the survivor's NEW state
  float score_rev(int o, const float* lb, int cls_b, int size_b, float s_, int* mc) const {
    float cdl = 0.0f;
    int m = ocls[o];
    if (ocls[o] != cls_b) {
      const float* la = &lp[(size_t)o * C];
      int best = 0;
      float bestv = la[0] + lb[0];
      for (int c = 1; c < C; c++) { const float v = la[c] + lb[c]; if (v > bestv) { bestv = v; best = c; } }
      m = best;
      cdl = bestv - la[ocls[o]] - lb[cls_b];
    }
    *mc = m;
    const size_t den = (size_t)osize[o] + (size_t)size_b;
    return (s_ * omf + cdl) / den + bias;
  }
  void apply_plan(const Plan& P) {
    steps++;
    if (!P.merging) { store(P.r, P.f); return; }
    merges++;
    const int a = P.a, b = P.b, r = P.r;
    q.erase(P.key); if (front_m) front_remove(P.key);
    prio[r] = -1.0f; r2[r] = -1;
    ocls[a] = P.mc; osize[a] = P.newsize;
    for (int c = 0; c < C; c++) lp[(size_t)a * C + c] = P.newlp[c];
    adj[a].erase(b); adj[b].erase(a);
    for (const PlanItem& it : P.items) {
      adj[it.c3].erase(b);
      if (it.fold) {
        S[it.tr] = it.newS;
        if (prio[it.t] >= 0.0f) { q.erase(Key{prio[it.t], it.t}); if (front_m) front_remove(Key{prio[it.t], it.t}); }
        prio[it.t] = -1.0f; r2[it.t] = -1;
      } else {
        r1[it.t] = std::min(a, it.c3); r2[it.t] = std::max(a, it.c3);
        adj[a][it.c3] = it.t; adj[it.c3][a] = it.t;
      }
      store(it.tr, it.newprio);
    }
    std::unordered_map<int, int>().swap(adj[b]);
    parent[b] = a;
  }
  void run_executor() {
    std::vector<long long> wmark(N, 0), rmark(N, 0);
    std::vector<Key> win;
    std::vector<Plan> plans;
    long long round = 0;
    while (!q.empty()) {
      round++;
      win.clear();
      if (front_m) {
        if ((int)front.size() < executor_k && front.size() < q.size()) front_rebuild();
        for (int i = 0; i < executor_k && i < (int)front.size(); i++) win.push_back(front[i]);
        // (the check: the front's first entries are the queue's)
        auto it = q.begin();
        fr_checks++;
        for (size_t i = 0; i < win.size(); ++i, ++it)
          if (it == q.end() || it->r != win[i].r || it->p != win[i].p) { fr_errors++; break; }
        if (win.size() < (size_t)executor_k && win.size() != q.size()) fr_errors++;
      } else {
        auto it = q.begin(); for (int i = 0; i < executor_k && it != q.end(); ++i, ++it) win.push_back(*it);
      }
      plans.resize(win.size());
      for (size_t i = 0; i < win.size(); i++) make_plan(win[i], plans[i]);       // all from the round's start state
      bool has_best = false; Key best{0.0f, 0};
      for (size_t i = 0; i < win.size(); i++) {
        const Plan& P = plans[i];
        if (i > 0) {
          if (has_best && best < win[i]) break;                      // a committed pop produced an entry that comes first
          bool clash = false;
          if (P.merging) {
            clash = wmark[P.a] == round || rmark[P.a] == round || wmark[P.b] == round || rmark[P.b] == round;
            for (const PlanItem& it : P.items) if (wmark[it.c3] == round) { clash = true; break; }
          } else {
            clash = wmark[P.a] == round || wmark[P.b] == round;
          }
          if (clash) break;
        }
        if (P.merging) {
          wmark[P.a] = rmark[P.a] = wmark[P.b] = rmark[P.b] = round;
          for (const PlanItem& it : P.items) rmark[it.c3] = round;
        } else {
          rmark[P.a] = round; rmark[P.b] = round;
        }
        if (P.has_best && (!has_best || P.best < best)) { best = P.best; has_best = true; }
        apply_plan(P);
      }
    }
    ex_rounds = round;
  }

  void run(const float* cls_p, const float* same_p, const int* offs) {
    N = W * H;
    lp.resize((size_t)N * C); ocls.resize(N); osize.assign(N, 1); parent.resize(N); adj.resize(N);
    stamp.assign(N, 0);
    if (rw) rstamp.assign(N, 0);
    if (parallel_study) batch_mark.assign(N, 0);
    if (rounds_k) { rs_wmark.assign(N, 0); rs_rmark.assign(N, 0); }
    acap.assign(N, cap0); alen.assign(N, 0); bump = (long long)N * cap0;
    for (int p = 0; p < N; p++) {
      float* l = &lp[(size_t)p * C];
      int best = 0;
      for (int c = 0; c < C; c++) { l[c] = logf(cls_p[(size_t)c * N + p]); if (l[c] > l[best]) best = c; }
      ocls[p] = best; parent[p] = p;
    }
    const size_t NL = (size_t)N * O;
    r1.assign(NL, -1); r2.assign(NL, -1); S.assign(NL, 0.0f); prio.assign(NL, -1.0f);
    if (rekey_study) rekeyed.assign(NL, 0);
    for (int p = 0; p < N; p++) {
      const int row = p / W, col = p % W;
      for (int k = 0; k < O; k++) {
        const int rr = row + offs[2 * k], cc = col + offs[2 * k + 1];
        if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
        const int qx = rr * W + cc;
        const float sp = same_p[(size_t)k * N + p];
        const float diff = (float)log(1.0 - (double)sp);
        const float same = logf(sp);
        const int r = p * O + k;
        live_records++;
        r1[r] = std::min(p, qx); r2[r] = std::max(p, qx);
        S[r] = same - diff;
        adj[p][qx] = r; adj[qx][p] = r;
        alen[p]++; alen[qx]++;
        int mc;
        const float f = score(r1[r], r2[r], S[r], &mc);
        prio[r] = f;
        if (f >= 0.0f) q.insert(Key{f, r});
      }
    }
    if (executor_k) { run_executor(); return; }
    long long ev = 0;
    n_start = live_records;
    float run_min = 3.0e38f;
    while (!q.empty()) {
      const Key top = *q.begin();
      const int r = top.r;
      // tied: a second live record with the bit-equal stored priority
      bool tied = false;
      { auto it = q.begin(); ++it; if (it != q.end() && it->p == top.p) tied = true; }
      steps++; ev++;
      if (rounds_k) rs_begin(top);
      if (parallel_study) {
        if (top.p <= run_min) { close_super_event(); n_roots++; run_min = top.p; }
        cur_root_events++;
      }
      tied_steps += tied;
      if (track) {
        // suffix minima of the popped priorities (non-strict): entries above the new one leave
        while (!stk.empty() && stk.back().first > top.p) {
          if (stk.size() >= 2 && stk[stk.size() - 2].first == stk.back().first) tied_pairs_on_stack--;
          stk.pop_back();
        }
        if (!stk.empty() && stk.back().first == top.p) tied_pairs_on_stack++;
        stk.push_back(std::make_pair(top.p, ev));
        if ((long long)stk.size() > max_depth) max_depth = (long long)stk.size();
      }
      const int x = r1[r], y = r2[r];
      int mc;
      const float f = score(x, y, S[r], &mc);
      const bool merging = f == top.p;
      touch(x, ev, merging); touch(y, ev, merging);
      if (parallel_study) { cur_fp.push_back(x); cur_fp.push_back(y); }
      if (rounds_k) { (merging ? rs_w : rs_r).push_back(x); (merging ? rs_w : rs_r).push_back(y); }
      if (!merging) { store(r, f); last_refreshed = r; if (rounds_k) rs_end(); continue; }
      if (last_refreshed == r) repop_merges++;
      last_refreshed = -1;
      // merge
      int a = x, b = y;
      if (osize[a] < osize[b]) std::swap(a, b);
      merges++; tied_merges += tied;
      q.erase(top); prio[r] = -1.0f; r2[r] = -1;
      live_records--;
      if (rekey_study && rekeyed[r]) live_rekeyed--;
      ocls[a] = mc;
      osize[a] += osize[b];
      float* la = &lp[(size_t)a * C];
      const float* lb = &lp[(size_t)b * C];
      for (int c = 0; c < C; c++) la[c] += lb[c];
      adj[a].erase(b); adj[b].erase(a);
      {
        // the engine: room for every record the survivor may adopt, checked before the walk
        const int la = alen[a], lb = alen[b];
        if (la + lb > acap[a]) {
          int need = growth_pct ? (int)((long long)(la + lb) * growth_pct / 100) : growth * (la + lb);
          int nc = reuse ? 1 : 0;
          if (reuse == 1) { nc = round_to; while (nc < need) nc <<= 1; } else nc = ((need + round_to - 1) / round_to) * round_to;
          alloc_block(nc);
          free_block(acap[a]);
          reallocs++; moved += (long long)adj[a].size();
          acap[a] = nc; alen[a] = (int)adj[a].size();       // (dead entries are dropped on the way)
        }
      }
      int adopted_n = 0;
      { int bkt = 0; while ((1u << (bkt + 1)) <= adj[b].size() && bkt < 15) bkt++; walk_hist[bkt]++; if ((long long)adj[b].size() > walk_max) walk_max = (long long)adj[b].size(); }
      for (auto& kv : adj[b]) {
        const int c3 = kv.first, t = kv.second;
        touch(c3, ev, false);
        if (parallel_study) cur_fp.push_back(c3);
        if (rounds_k) rs_r.push_back(c3);
        adj[c3].erase(b);
        // a record modified (or retired) while its stored priority equals that of an event on the stack -- an
        // ancestor of this event or the event itself: it is a tied sibling whose turn might have come first
        auto hit = adj[a].find(c3);
        int tr;
        if (hit != adj[a].end()) {
          tr = hit->second;
          sibling_check(prio[t]);
          sibling_check(prio[tr]);
          S[tr] += S[t];
          if (prio[t] >= 0.0f) q.erase(Key{prio[t], t});
          prio[t] = -1.0f; r2[t] = -1;
          live_records--;
          if (rekey_study && rekeyed[t]) live_rekeyed--;
        } else {
          tr = t;
          if (rekey_study && !rekeyed[t]) {
            rekeyed[t] = 1;
            if (++live_rekeyed > peak_rekeyed) { peak_rekeyed = live_rekeyed; live_at_peak = live_records; }
          }
          sibling_check(prio[t]);
          r1[t] = std::min(a, c3); r2[t] = std::max(a, c3);
          adj[a][c3] = t; adj[c3][a] = t;
          adopted_n++;
        }
        int m2;
        store(tr, score(r1[tr], r2[tr], S[tr], &m2));
      }
      alen[a] += adopted_n;
      free_block(acap[b]);
      std::unordered_map<int, int>().swap(adj[b]);
      parent[b] = a;
      if (rounds_k) rs_end();
    }
  }

  void finish_study() {
    if (!parallel_study) return;
    close_super_event();
    int b = 0; while ((1LL << (b + 1)) <= batch_len && b < 15) b++;
    batch_hist[b]++; n_batches++;
  }
  // Rounds of an executor that looks at the next K super-events in order and runs every one whose footprint is
  // disjoint from ALL earlier ones in its window (run or not); the others wait for the next round.
  void window_study(int K) {
    std::vector<long long> mark(N, 0);
    std::vector<size_t> pending;
    size_t next = 0;
    long long rounds = 0, done = 0;
    while (next < all_fp.size() || !pending.empty()) {
      while (pending.size() < (size_t)K && next < all_fp.size()) pending.push_back(next++);
      rounds++;
      std::vector<size_t> left;
      for (size_t i : pending) {
        bool clash = false;
        for (int o : all_fp[i]) if (mark[o] == rounds) { clash = true; break; }
        for (int o : all_fp[i]) mark[o] = rounds;
        if (clash) left.push_back(i); else done++;
      }
      pending.swap(left);
    }
    fprintf(stderr, "window study K = %d: %lld rounds for %lld super-events = %.1f per round\n", K, rounds, done, (double)done / (double)(rounds ? rounds : 1));
  }
  int root(int p) { while (parent[p] != p) p = parent[p]; return p; }
};

}  // namespace

extern "C" int exact_model_run(const float* class_pred, const float* adj_pred, int C, int O, int W, int H,
                               const int* offs, float omf, float bias, int track, int* partition, int* obj_class,
                               double* stats) {
  Model m;
  if (const char* e = getenv("XM_CAP0")) m.cap0 = atoi(e);
  if (const char* e = getenv("XM_GROWTH")) m.growth = atoi(e);
  if (const char* e = getenv("XM_ROUND")) m.round_to = atoi(e);
  if (const char* e = getenv("XM_REUSE")) m.reuse = atoi(e);
  if (const char* e = getenv("XM_GROWTH_PCT")) m.growth_pct = atoi(e);
  if (const char* e = getenv("XM_PARALLEL")) m.parallel_study = atoi(e);
  if (const char* e = getenv("XM_REKEY")) m.rekey_study = atoi(e);
  if (const char* e = getenv("XM_RW")) m.rw = atoi(e);
  if (const char* e = getenv("XM_ROUNDS")) m.rounds_k = atoi(e);
  if (const char* e = getenv("XM_EXECUTOR")) m.executor_k = atoi(e);
  if (const char* e = getenv("XM_FRONT")) m.front_m = atoi(e);
  m.C = C; m.O = O; m.W = W; m.H = H; m.omf = omf; m.bias = bias; m.track = track;
  m.run(class_pred, adj_pred, offs);
  m.finish_study();
  if (m.parallel_study) {
    fprintf(stderr, "parallel study: %lld pops, %lld root super-events (largest %lld pops), %lld batches of consecutive disjoint super-events: mean %.1f super-events = %.1f pops per batch; batch lengths by log2 bucket:", m.steps, m.n_roots, m.max_root_events, m.n_batches, (double)m.n_roots / (double)(m.n_batches ? m.n_batches : 1), (double)m.steps / (double)(m.n_batches ? m.n_batches : 1));
    for (int i = 0; i < 16; i++) fprintf(stderr, " %lld", m.batch_hist[i]);
    fprintf(stderr, "\n");
    if (m.parallel_study >= 2) { m.window_study(16); m.window_study(64); m.window_study(256); }
  }
  if (m.executor_k)
    fprintf(stderr, "prefix executor run, window %d: %lld pops, %lld merges in %lld rounds = %.2f pops per round\n", m.executor_k,
            m.steps, m.merges, m.ex_rounds, (double)m.steps / (double)(m.ex_rounds ? m.ex_rounds : 1));
  if (m.executor_k && m.front_m)
    fprintf(stderr, "sorted front of %d entries: %lld rebuilds (one per %.0f pops), %lld inserts, %lld window checks, %lld WRONG windows\n",
            m.front_m, m.fr_rebuilds, (double)m.steps / (double)(m.fr_rebuilds ? m.fr_rebuilds : 1), m.fr_inserts, m.fr_checks, m.fr_errors);
  if (m.rounds_k) {
    if (m.rs_in_round > 0) m.rs_rounds++;
    fprintf(stderr, "prefix executor, window %d: %lld pops in %lld rounds = %.2f per round; a round ended because the next pop was not the next entry of the window %lld, the window was used up %lld, footprints clashed %lld; rounds by log2 of their length:",
            m.rounds_k, m.steps, m.rs_rounds, (double)m.steps / (double)(m.rs_rounds ? m.rs_rounds : 1), m.rs_cut_order, m.rs_cut_window, m.rs_cut_clash);
    for (int i = 0; i < 12; i++) fprintf(stderr, " %lld", m.rs_hist[i]);
    fprintf(stderr, "\n");
  }
  if (getenv("XM_WALKS")) {
    fprintf(stderr, "records of the absorbed object at a merge (longest %lld), by log2 bucket:", m.walk_max);
    for (int i = 0; i < 16; i++) fprintf(stderr, " %lld", m.walk_hist[i]);
    fprintf(stderr, "\n");
  }
  if (getenv("XM_KINDS"))
    fprintf(stderr, "conflicts by kind: write after write %lld, read after write %lld, write after read %lld, rival record %lld\n",
            m.kind_count[0], m.kind_count[1], m.kind_count[2], m.kind_count[3]);
  if (m.rekey_study)
    fprintf(stderr, "re-key study: %lld records at the start; at most %lld live records carried a key other than their pixel pair (%.1f %% of the start; %lld records were live then)\n",
            m.n_start, m.peak_rekeyed, 100.0 * (double)m.peak_rekeyed / (double)(m.n_start ? m.n_start : 1), m.live_at_peak);
  for (int p = 0; p < W * H; p++) { const int o = m.root(p); partition[p] = o; obj_class[p] = m.ocls[o]; }
  if (stats) {
    stats[0] = (double)m.steps; stats[1] = (double)m.merges; stats[2] = (double)m.tied_steps;
    stats[3] = (double)m.tied_merges; stats[4] = (double)m.tied_conflicts; stats[5] = (double)m.max_depth;
    stats[6] = (double)m.bump; stats[7] = (double)m.reallocs; stats[8] = (double)m.repop_merges;
    stats[9] = (double)m.fr_errors; stats[10] = (double)m.fr_rebuilds; stats[11] = (double)m.ex_rounds;
  }
  return 0;
}
