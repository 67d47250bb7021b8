// pysegmenter_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference PYTHON merger (variant "pysegmenter",
// utils/segmenter.py), float64 throughout (the reference is only usable at size with
// float64 inputs: its own assertion at segmenter.py:520 trips on float32 drift).
//
// Parity status: PINNED by golden vectors produced in the build container by importing
// the reference's utils.segmenter.ObjectSegmenter on float64 inputs
// (tests/golden/make_golden.py writes tests/golden/py_*.npz).
//
// It differs from the C++ variant in every row of SURVEY.md Appendix A.3:
//   priority = (oml*omf + cdl + bias) / (n1*n2)          segmenter.py:189-193
//   merge iff re-scored priority >= popped priority        segmenter.py:470
//   min-heap of (-priority, record) with Python tuple ordering: ties on the key fall
//     through to record.__eq__ (same id pair) and record.__lt__ (CURRENT priorities)
//                                                          segmenter.py:203-218, 289, 465
//   tombstone priority -100000.0                           segmenter.py:562
//   records of deleted objects stay scoreable; merge() rejects them  segmenter.py:513-514
//   per-object adjacency "dict": iteration = insertion order with deletions; an adopted
//     record is re-appended at the end of both lists       segmenter.py:541-574
//   prune(200) after the loop                              segmenter.py:351-375, 478
//   labels in ascending surviving id                       segmenter.py:377-389
// heapq's sift routines are restated from CPython's documented algorithm (heapq.py).

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <list>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct PySeg {
  int C, O, W, H, N;
  double sdb, omf, bias;
  const double* cls_p;
  double* same_p;
  const int* offs;

  std::vector<double> lp;        // [N*C]
  std::vector<int> ocls, osize;
  std::vector<double> osame;
  std::vector<char> alive;
  std::vector<int> parent;
  // ordered adjacency "dict": list of record ids in insertion order + lookup by other id
  struct Slot { int rec; std::list<int>::iterator it; };
  std::vector<std::list<int> > order;
  std::vector<std::unordered_map<int, Slot> > lookup;

  std::vector<int> r1, r2;
  std::vector<double> rsame, rdiff, roml, rcdl, rprio;
  std::vector<int> rmcls;

  std::vector<std::pair<double, int> > heap;   // (-priority at push time, record)
  long long n_pops = 0, n_merges = 0;
  int error = 0;

  // ---- Python tuple "<" on (key, record) ----
  bool lt(const std::pair<double, int>& x, const std::pair<double, int>& y) const {
    if (x.first != y.first) return x.first < y.first;
    const int a = x.second, b = y.second;
    if (r1[a] == r1[b] && r2[a] == r2[b]) return false;   // records compare equal
    return rprio[a] < rprio[b];
  }
  void siftdown(size_t startpos, size_t pos) {
    std::pair<double, int> item = heap[pos];
    while (pos > startpos) {
      size_t parentpos = (pos - 1) >> 1;
      if (lt(item, heap[parentpos])) { heap[pos] = heap[parentpos]; pos = parentpos; continue; }
      break;
    }
    heap[pos] = item;
  }
  void siftup(size_t pos) {
    const size_t endpos = heap.size(), startpos = pos;
    std::pair<double, int> item = heap[pos];
    size_t child = 2 * pos + 1;
    while (child < endpos) {
      size_t right = child + 1;
      if (right < endpos && !lt(heap[child], heap[right])) child = right;
      heap[pos] = heap[child];
      pos = child;
      child = 2 * pos + 1;
    }
    heap[pos] = item;
    siftdown(startpos, pos);
  }
  void push(double prio, int r) {
    heap.push_back(std::make_pair(-prio, r));
    siftdown(0, heap.size() - 1);
  }
  std::pair<double, int> pop() {
    std::pair<double, int> last = heap.back();
    heap.pop_back();
    if (!heap.empty()) {
      std::pair<double, int> top = heap[0];
      heap[0] = last;
      siftup(0);
      return top;
    }
    return last;
  }

  void sort_ids(int r) { if (r1[r] > r2[r]) std::swap(r1[r], r2[r]); }

  void rescore(int r) {
    const int a = r1[r], b = r2[r];
    if (ocls[a] == ocls[b]) {
      rcdl[r] = 0.0;
      rmcls[r] = ocls[a];
    } else {
      const double* la = &lp[(size_t)a * C];
      const double* lb = &lp[(size_t)b * C];
      int best = 0;
      double bestv = la[0] + lb[0];
      for (int c = 1; c < C; c++) {
        double v = la[c] + lb[c];
        if (v > bestv) { bestv = v; best = c; }
      }
      rmcls[r] = best;
      rcdl[r] = bestv - la[ocls[a]] - lb[ocls[b]];
    }
    const double den = (double)((long long)osize[a] * (long long)osize[b]);
    rprio[r] = (roml[r] * omf + rcdl[r] + bias) / den;
  }

  void dict_set(int obj, int other, int r) {        // insert new key at the end
    order[obj].push_back(r);
    Slot s; s.rec = r; s.it = --order[obj].end();
    lookup[obj][other] = s;
  }
  void dict_del(int obj, int other) {
    std::unordered_map<int, Slot>::iterator f = lookup[obj].find(other);
    if (f == lookup[obj].end()) { error = -20; return; }   // Python would raise KeyError
    order[obj].erase(f->second.it);
    lookup[obj].erase(f);
  }

  void build() {
    N = W * H;
    if (sdb != 0.0) {
      for (size_t i = 0; i < (size_t)O * N; i++) {
        double logit = std::log(same_p[i]) - std::log(1.0 - same_p[i]) + sdb;
        same_p[i] = 1.0 / (1.0 + std::exp(-logit));
      }
    }
    lp.assign((size_t)N * C, 0.0);
    ocls.resize(N); osize.assign(N, 1); osame.assign(N, 0.0); alive.assign(N, 1);
    parent.resize(N); order.resize(N); lookup.resize(N);
    for (int p = 0; p < N; p++) {
      double* l = &lp[(size_t)p * C];
      int best = 0;
      for (int c = 0; c < C; c++) {
        l[c] += std::log(cls_p[(size_t)c * N + p]);
        if (l[c] > l[best]) best = c;
      }
      ocls[p] = best;
      parent[p] = p;
    }
    for (int row = 0; row < H; row++)
      for (int col = 0; col < W; col++) {
        const int p = row * W + col;
        for (int k = 0; k < O; k++) {
          const int rr = row + offs[2 * k], cc = col + offs[2 * k + 1];
          if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
          const int q = rr * W + cc;
          const double sp = same_p[(size_t)k * N + p];
          const int r = (int)r1.size();
          r1.push_back(p); r2.push_back(q);
          const double ls = std::log(sp), ld = std::log(1.0 - sp);
          rdiff.push_back(ld); rsame.push_back(ls); roml.push_back(ls - ld);
          rcdl.push_back(0); rprio.push_back(0); rmcls.push_back(0);
          sort_ids(r);
          rescore(r);
          dict_set(p, q, r);
          dict_set(q, p, r);
          if (rprio[r] >= 0) push(rprio[r], r);
        }
      }
  }

  void merge(int r) {
    int a = r1[r], b = r2[r];
    if (!alive[a] || !alive[b]) return;
    if (a == b) return;
    if (osize[b] > osize[a]) std::swap(a, b);
    if (std::fabs(roml[r] - (rsame[r] - rdiff[r])) >= 0.001) { error = -21; return; }  // assert :520
    ocls[a] = rmcls[r];
    osize[a] += osize[b];
    double* la = &lp[(size_t)a * C];
    const double* lb = &lp[(size_t)b * C];
    for (int c = 0; c < C; c++) la[c] += lb[c];
    osame[a] += rsame[r] + osame[b];
    dict_del(a, b);
    dict_del(b, a);
    if (error) return;
    // iterate b's dict in insertion order (nothing in this loop edits b's dict)
    for (std::list<int>::iterator it = order[b].begin(); it != order[b].end(); ++it) {
      const int t = *it;
      const int c3 = (r1[t] == b) ? r2[t] : r1[t];
      dict_del(c3, b);
      if (error) return;
      if (r1[t] == b) r1[t] = a;
      if (r2[t] == b) r2[t] = a;
      sort_ids(t);
      std::unordered_map<int, Slot>::iterator hit = lookup[a].find(c3);
      if (hit != lookup[a].end()) {
        const int u = hit->second.rec;
        roml[u] += roml[t];
        rdiff[u] += rdiff[t];
        rsame[u] += rsame[t];
        rprio[t] = -100000.0;
        rescore(u);
        if (rprio[u] >= 0) push(rprio[u], u);
      } else {
        dict_set(a, c3, t);
        dict_set(c3, a, t);
        rescore(t);
        if (rprio[t] >= 0) push(rprio[t], t);
      }
    }
    alive[b] = 0;
    parent[b] = a;
    n_merges++;
  }

  void run() {
    while (!heap.empty() && !error) {
      std::pair<double, int> e = pop();
      n_pops++;
      const double q = -e.first;
      const int r = e.second;
      if (q != rprio[r]) continue;
      rescore(r);      // also for records of deleted objects, as the reference does
      if (rprio[r] >= q) merge(r);
      else if (rprio[r] >= 0) push(rprio[r], r);
    }
  }

  int root(int p) {
    int x = p;
    while (parent[x] != x) x = parent[x];
    while (parent[p] != x) { int nx = parent[p]; parent[p] = x; p = nx; }
    return x;
  }

  // prune(threshold) + output_mask; returns <0 where the reference raises NameError
  int emit(double threshold, long long* output, int* object_class, int* partition) {
    std::vector<int> objs;
    for (int o = 0; o < N; o++) if (alive[o]) objs.push_back(o);
    for (int p = 0; p < N; p++) if (partition) partition[p] = root(p);
    int background = -1, best = 0;
    for (size_t i = 0; i < objs.size(); i++)
      if (ocls[objs[i]] == 0 && osize[objs[i]] > best) { background = objs[i]; best = osize[objs[i]]; }
    std::vector<char> pruned(N, 0);
    for (size_t i = 0; i < objs.size(); i++) {
      const int o = objs[i];
      const double score = lp[(size_t)o * C + ocls[o]] - lp[(size_t)o * C];
      if (score < threshold) {
        if (background < 0) return -10;        // NameError: background_obj unbound
        if (o != background) pruned[o] = 1;
      }
    }
    std::vector<int> label(N, 0);
    int k = 1;
    for (int i = 0; i < N; i++) object_class[i] = -1;
    for (size_t i = 0; i < objs.size(); i++) {
      const int o = objs[i];
      if (pruned[o] || ocls[o] == 0) continue;
      object_class[k - 1] = ocls[o];
      label[o] = k++;
    }
    for (int p = 0; p < N; p++) output[p] = label[root(p)];
    return 0;
  }

  double total_logprob() {
    double cls_term = 0, same_term = 0, diff_term = 0;
    for (int p = 0; p < N; p++) cls_term += std::log(cls_p[(size_t)ocls[root(p)] * N + p]);
    for (int row = 0; row < H; row++)
      for (int col = 0; col < W; col++) {
        const int p = row * W + col, op = root(p);
        for (int k = 0; k < O; k++) {
          const int rr = row + offs[2 * k], cc = col + offs[2 * k + 1];
          if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
          const double sp = same_p[(size_t)k * N + p];
          if (root(rr * W + cc) == op) same_term += std::log(sp);
          else diff_term += std::log(1.0 - sp);
        }
      }
    return cls_term + (diff_term + same_term) * omf;
  }
};

}  // namespace

extern "C" {

// Inputs are float64, already clipped to [eps32, 1-eps32] (segmenter.py:232-234); same_probs is
// rewritten in place when same_different_bias != 0.  output is int64 [H*W] (numpy "int").
// stats[6] = {total_logprob (pre-prune partition), n_objects_before_prune, n_pops, n_merges,
//             n_initial_records, 0}.
// Returns 0; -10 where the reference raises NameError (no class-0 object to prune into);
// -20/-21 where it would raise KeyError/AssertionError; -1..-5 for invalid arguments.
int oracle_pysegmenter_run(const double* class_probs, double* same_probs, int num_classes,
                           int offset_dim, int img_height, int img_width, const int* offset_list,
                           double same_different_bias, double object_merge_factor,
                           double merge_logprob_bias, double prune_threshold,
                           long long* output, int* object_class, int* partition, double* stats) {
  if (!class_probs || !same_probs || !offset_list || !output || !object_class) return -1;
  if (img_width <= 0 || img_height <= 0 || num_classes <= 0 || offset_dim < 0) return -2;
  for (int a = 0; a < offset_dim; a++)
    for (int b = 0; b < offset_dim; b++) {
      const bool neg = offset_list[2 * a] == -offset_list[2 * b] &&
                       offset_list[2 * a + 1] == -offset_list[2 * b + 1];
      const bool dup = a != b && offset_list[2 * a] == offset_list[2 * b] &&
                       offset_list[2 * a + 1] == offset_list[2 * b + 1];
      if (neg || dup) return -5;
    }
  PySeg s;
  s.C = num_classes; s.O = offset_dim; s.W = img_width; s.H = img_height;
  s.sdb = same_different_bias; s.omf = object_merge_factor; s.bias = merge_logprob_bias;
  s.cls_p = class_probs; s.same_p = same_probs; s.offs = offset_list;
  s.build();
  const size_t n_rec = s.r1.size();
  s.run();
  if (s.error) return s.error;
  int n_obj = 0;
  for (int o = 0; o < s.N; o++) n_obj += s.alive[o];
  if (stats) {
    stats[0] = s.total_logprob();
    stats[1] = n_obj;
    stats[2] = (double)s.n_pops;
    stats[3] = (double)s.n_merges;
    stats[4] = (double)n_rec;
    stats[5] = 0;
  }
  return s.emit(prune_threshold, output, object_class, partition);
}

}  // extern "C"
