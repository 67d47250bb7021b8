// proto_exact_rounds.cpp -- CPU prototype (test tooling, not product code) of the EXACT parallel
// schedule used by the GPU "exact rounds" engine (mergenet_amd/csrc/mn_kernels_exact.h).
//
// The reference merger (utils/csegment/segment.cc:539-573, Merge :602-727) pops one record at a
// time from a priority queue.  This prototype executes the SAME events (pop -> re-score -> merge
// or re-queue) in parallel rounds and checks that the result equals the sequential order.
//
//   virtual time   tau(record) = min(stored priority, tau of the event that stored it); the
//                  sequential order is descending tau (then descending stored priority): an
//                  event with larger tau always precedes one with smaller tau.
//   ready rule     record r = (a,b) may execute in a round iff it is the best record (by
//                  (tau, stored, -u, -v)) of both a and b and -- if it merges -- no neighbour of
//                  the absorbed object has a better record than r.
//   two phases     all ready events read the state of the round's start, then all commit.
//   certificate    every object keeps the tau of the last event that wrote / read it; an event
//                  that would have to go BEFORE one already executed on the same object is an
//                  order violation (counted; zero = the schedule is a reordering of the
//                  sequential execution by swaps of commuting events).
//
// Usage: proto_exact_rounds C O W H class.f32 same.f32 offs.i32 sdb omf bias  -> prints stats
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <unordered_map>
#include <vector>

struct Key {
  float tau, q; int u, v;
};
static inline bool better(const Key& a, const Key& b) {   // a before b
  if (a.tau != b.tau) return a.tau > b.tau;
  if (a.q != b.q) return a.q > b.q;
  if (a.u != b.u) return a.u < b.u;
  return a.v < b.v;
}

struct Eng {
  int C, O, W, H, N;
  float omf, bias;
  std::vector<float> lp; std::vector<int> ocls, osize, parent;
  std::vector<std::unordered_map<int, int>> adj;   // neighbour object -> record
  std::vector<int> ru, rv; std::vector<float> rS, rst, rtau; std::vector<char> alive;
  long long n_events = 0, n_merges = 0, n_rounds = 0, n_viol = 0; FILE* evlog = nullptr;
  void logev(int u, int v, float q, float f, float tau) { if (evlog) fprintf(evlog, "%d %d %.9g %.9g %.9g %lld\n", u, v, q, f, tau, n_rounds); }

  float score(int a, int b, float S, int* mc) {
    float cdl = 0; int m = ocls[a];
    if (ocls[a] != ocls[b]) {
      const float* la = &lp[(size_t)a * C]; const float* lb = &lp[(size_t)b * C];
      int best = 0; float bv = la[0] + lb[0];
      for (int c = 1; c < C; c++) { float v = la[c] + lb[c]; if (v > bv) { bv = v; best = c; } }
      m = best; cdl = bv - la[ocls[a]] - lb[ocls[b]];
    }
    *mc = m;
    size_t den = (size_t)osize[a] + (size_t)osize[b];
    return (S * omf + cdl) / den + bias;
  }

  void build(const float* cls_p, const float* same_p, const int* offs) {
    N = W * H;
    lp.resize((size_t)N * C); ocls.resize(N); osize.assign(N, 1); parent.resize(N); adj.resize(N);
    for (int p = 0; p < N; p++) {
      int best = 0;
      for (int c = 0; c < C; c++) { lp[(size_t)p * C + c] = logf(cls_p[(size_t)c * N + p]); if (lp[(size_t)p * C + c] > lp[(size_t)p * C + best]) best = c; }
      ocls[p] = best; parent[p] = p;
    }
    for (int row = 0; row < H; row++) for (int col = 0; col < W; col++) {
      int p = row * W + col;
      for (int k = 0; k < O; k++) {
        int rr = row + offs[2 * k], cc = col + offs[2 * k + 1];
        if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
        int q = rr * W + cc; float sp = same_p[(size_t)k * N + p];
        float S = logf(sp) - (float)log(1.0 - sp);
        int r = (int)ru.size();
        ru.push_back(std::min(p, q)); rv.push_back(std::max(p, q)); rS.push_back(S);
        int mc; float pr = score(ru[r], rv[r], S, &mc);
        rst.push_back(pr); rtau.push_back(pr); alive.push_back(1);
        adj[p][q] = r; adj[q][p] = r;
      }
    }
  }

  Key keyof(int r) const { Key k; k.tau = rtau[r]; k.q = rst[r]; k.u = ru[r]; k.v = rv[r]; return k; }

  // ---- sequential execution with the same arithmetic (ties: (u,v)) -------------------------
  void run_sequential() {
    auto cmp = [&](const std::pair<Key, int>& x, const std::pair<Key, int>& y) { return better(y.first, x.first); };
    std::priority_queue<std::pair<Key, int>, std::vector<std::pair<Key, int>>, decltype(cmp)> pq(cmp);
    for (int r = 0; r < (int)ru.size(); r++) if (rst[r] >= 0) { Key k = keyof(r); k.tau = k.q; pq.push({k, r}); }
    while (!pq.empty()) {
      auto top = pq.top(); pq.pop();
      int r = top.second;
      if (!alive[r] || top.first.q != rst[r] || top.first.u != ru[r] || top.first.v != rv[r]) continue;
      n_events++;
      int mc; float f = score(ru[r], rv[r], rS[r], &mc);
      logev(ru[r], rv[r], rst[r], f, 0);
      if (f != rst[r]) { rst[r] = f; if (f >= 0) { Key k = keyof(r); k.tau = k.q; pq.push({k, r}); } continue; }
      int a = ru[r], b = rv[r];
      if (osize[a] < osize[b]) std::swap(a, b);
      std::vector<int> touched;
      do_merge(r, a, b, mc, touched);
      for (int t : touched) if (rst[t] >= 0) { Key k = keyof(t); k.tau = k.q; pq.push({k, t}); }
    }
  }

  void do_merge(int r, int a, int b, int mc, std::vector<int>& touched) {
    ocls[a] = mc; osize[a] += osize[b];
    for (int c = 0; c < C; c++) lp[(size_t)a * C + c] += lp[(size_t)b * C + c];
    adj[a].erase(b); adj[b].erase(a); alive[r] = 0; parent[b] = a; n_merges++;
    for (auto& kv : adj[b]) {
      int x = kv.first, t = kv.second;
      adj[x].erase(b);
      auto hit = adj[a].find(x);
      int tgt;
      if (hit != adj[a].end()) { tgt = hit->second; rS[tgt] += rS[t]; alive[t] = 0; }
      else { tgt = t; ru[t] = std::min(a, x); rv[t] = std::max(a, x); adj[a][x] = t; adj[x][a] = t; }
      int m2; rst[tgt] = score(ru[tgt], rv[tgt], rS[tgt], &m2);
      touched.push_back(tgt);
    }
    adj[b].clear();
  }

  // ---- parallel rounds ------------------------------------------------------------------------
  struct Plan { int r, a, b, mc; bool merge; float fresh; std::vector<std::pair<int, float>> newst; };

  void run_rounds(int strict) {
    std::vector<int> best(N, -1);
    std::vector<float> stampW(N, INFINITY), stampR(N, INFINITY);
    std::vector<char> live(N, 1);
    std::vector<int> liveobjs(N);
    for (int i = 0; i < N; i++) liveobjs[i] = i;
    auto recompute = [&](int x) {
      int bi = -1; Key bk{};
      for (auto& kv : adj[x]) { int t = kv.second; if (!(rst[t] >= 0)) continue; Key k = keyof(t); if (bi < 0 || better(k, bk)) { bi = t; bk = k; } }
      best[x] = bi;
    };
    for (int x = 0; x < N; x++) recompute(x);
    for (;;) {
      std::vector<int> ready;
      for (int a : liveobjs) {
        if (!live[a]) continue;
        int r = best[a]; if (r < 0) continue;
        int o = ru[r] == a ? rv[r] : ru[r];
        if (a != ru[r]) continue;            // the lower endpoint proposes
        if (best[o] != r) continue;
        int mc; float f = score(ru[r], rv[r], rS[r], &mc);
        bool ok = true;
        {
          int sa = ru[r], sb = rv[r];
          if (osize[sa] < osize[sb]) std::swap(sa, sb);
          const bool mrg = f == rst[r];
          Key kr = keyof(r);
          auto hot = [&](int x) { int bx = best[x]; return bx >= 0 && bx != r && better(keyof(bx), kr); };
          auto check = [&](int obj, int depth) {
            for (auto& kv : adj[obj]) { if (hot(kv.first)) { ok = false; return; }
              if (depth >= 2) for (auto& k2 : adj[kv.first]) { if (hot(k2.first)) { ok = false; return; }
                if (depth >= 3) for (auto& k3 : adj[k2.first]) if (hot(k3.first)) { ok = false; return; } } }
          };
          // strict 0: merges check the absorbed object's neighbours only; refreshes nothing
          if (strict == 0) { if (mrg) check(sb, 1); }
          else { check(sb, strict); if (ok) check(sa, strict); }
        }
        if (ok) ready.push_back(r);
      }
      if (ready.empty()) break;
      n_rounds++;
      std::sort(ready.begin(), ready.end(), [&](int x, int y) { return better(keyof(x), keyof(y)); });
      // phase 1: plan from the state at the start of the round
      std::vector<Plan> plans;
      for (int r : ready) {
        Plan P; P.r = r; int mc; P.fresh = score(ru[r], rv[r], rS[r], &mc); P.mc = mc; P.merge = P.fresh == rst[r];
        P.a = ru[r]; P.b = rv[r];
        if (P.merge) {
          if (osize[P.a] < osize[P.b]) std::swap(P.a, P.b);
          // new priorities of the absorbed object's records with the merged object's state
          int a = P.a, b = P.b;
          std::vector<float> lm(C);
          for (int c = 0; c < C; c++) lm[c] = lp[(size_t)a * C + c] + lp[(size_t)b * C + c];
          int nm = osize[a] + osize[b];
          for (auto& kv : adj[b]) {
            int x = kv.first, t = kv.second; if (t == r) continue;
            float S = rS[t]; auto hit = adj[a].find(x); if (hit != adj[a].end()) S = rS[hit->second] + rS[t];
            float cdl = 0;
            if (mc != ocls[x]) {
              const float* lx = &lp[(size_t)x * C]; float bv = lm[0] + lx[0]; for (int c = 1; c < C; c++) { float v = lm[c] + lx[c]; if (v > bv) bv = v; }
              // operand order of the reference: the record's obj1 is the lower id
              int lo = std::min(a, x);
              if (lo == a) cdl = bv - lm[mc] - lx[ocls[x]]; else cdl = bv - lx[ocls[x]] - lm[mc];
            }
            size_t den = (size_t)nm + (size_t)osize[x];
            P.newst.push_back({x, (S * omf + cdl) / den + bias});
          }
        }
        plans.push_back(P);
      }
      // phase 2: commit
      std::vector<int> dirty;
      for (auto& P : plans) {
        int r = P.r; n_events++;
        float tau_e = rtau[r];
        logev(ru[r], rv[r], rst[r], P.fresh, tau_e);
        if (getenv("DBGOBJ") && (ru[r] == 7585 && rv[r] == 7838)) { for (int o : {7585, 7838}) for (auto& kv : adj[o]) fprintf(stderr, "  obj %d nb %d rec st %.9g tau %.9g size %d\n", o, kv.first, rst[kv.second], rtau[kv.second], osize[kv.first]); }
        auto touchW = [&](int x) { if (tau_e > stampW[x] || tau_e > stampR[x]) n_viol++; stampW[x] = std::min(stampW[x], tau_e); };
        auto touchR = [&](int x) { if (tau_e > stampW[x]) n_viol++; stampR[x] = std::min(stampR[x], tau_e); };
        if (!P.merge) {
          touchR(ru[r]); touchR(rv[r]);
          rst[r] = P.fresh; rtau[r] = std::min(P.fresh, tau_e);
          dirty.push_back(ru[r]); dirty.push_back(rv[r]);
          continue;
        }
        int a = P.a, b = P.b;
        touchW(a); touchW(b);
        std::vector<std::pair<int, int>> nb(adj[b].begin(), adj[b].end());
        std::vector<int> touched;
        // the float sums of the survivor are formed exactly as in the sequential merge
        do_merge(r, a, b, P.mc, touched);
        live[b] = 0;
        // stored priorities: the planned values (computed from the round's start state)
        size_t j = 0;
        for (auto& kv : nb) {
          int x = kv.first; if (kv.second == r) continue;
          touchR(x);
          int t = adj[a][x];
          float planned = P.newst[j].second; (void)planned;
          // do_merge re-scored with the CURRENT state of x; in a two-phase round x may have been
          // changed by a colder event of this round, so the planned value is the one to keep
          if (P.newst[j].first != x) { fprintf(stderr, "plan order mismatch\n"); exit(2); }
          if (getenv("DBGPLAN") && rst[t] != P.newst[j].second) fprintf(stderr, "round %lld rec (%d,%d): do_merge %.9g planned %.9g\n", n_rounds, ru[t], rv[t], rst[t], P.newst[j].second);
          rst[t] = P.newst[j].second;
          rtau[t] = std::min(rst[t], tau_e);
          j++;
          dirty.push_back(x);
        }
        dirty.push_back(a);
      }
      for (int x : dirty) if (live[x]) recompute(x);
      if ((n_rounds & 63) == 0) { std::vector<int> nl; for (int x : liveobjs) if (live[x]) nl.push_back(x); liveobjs.swap(nl); }
    }
  }

  int root(int p) { while (parent[p] != p) p = parent[p]; return p; }
};

static std::vector<char> slurp(const char* path) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(1); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> b(n); if (fread(b.data(), 1, n, f) != (size_t)n) exit(1); fclose(f); return b;
}

int main(int argc, char** argv) {
  if (argc < 11) { fprintf(stderr, "usage\n"); return 1; }
  int C = atoi(argv[1]), O = atoi(argv[2]), W = atoi(argv[3]), H = atoi(argv[4]);
  auto cb = slurp(argv[5]), sb = slurp(argv[6]), ob = slurp(argv[7]);
  float omf = atof(argv[9]), bias = atof(argv[10]);
  int strict = argc > 11 ? atoi(argv[11]) : 0;
  Eng A, B;
  for (Eng* e : {&A, &B}) { e->C = C; e->O = O; e->W = W; e->H = H; e->omf = omf; e->bias = bias; e->build((const float*)cb.data(), (const float*)sb.data(), (const int*)ob.data()); }
  if (getenv("EVLOG")) { A.evlog = fopen("/tmp/proto/ev_seq.txt", "w"); B.evlog = fopen("/tmp/proto/ev_par.txt", "w"); }
  A.run_sequential();
  B.run_rounds(strict);
  int N = W * H; long long diff = 0;
  std::unordered_map<long long, int> pairs;
  { std::unordered_map<int,int> ab, ba;
    for (int p = 0; p < N; p++) { int x = A.root(p), y = B.root(p);
      auto i = ab.find(x); if (i == ab.end()) ab[x] = y; else if (i->second != y) { diff++; continue; }
      auto j = ba.find(y); if (j == ba.end()) ba[y] = x; else if (j->second != x) diff++; } }
  int na = 0, nb = 0; for (int p = 0; p < N; p++) { na += A.parent[p] == p; nb += B.parent[p] == p; }
  printf("sequential: events %lld merges %lld objects %d | rounds: events %lld merges %lld objects %d rounds %lld violations %lld | pixels with different root %lld\n",
         A.n_events, A.n_merges, na, B.n_events, B.n_merges, nb, B.n_rounds, B.n_viol, diff);
  return diff == 0 ? 0 : 3;
}
