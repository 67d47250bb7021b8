// csegment_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference C++ merger (variant "csegment"), used only as the
// checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
// under mergenet_amd/ may import, link or call it.
//
// Parity status: PINNED.  In the build container this file is checked bit-for-bit (mask
// labels, class list order, partition) against oracle/_ref/libcsegment_ref.so, which is
// the reference's own utils/csegment/segment.cc compiled unmodified (oracle/Makefile),
// on well-conditioned and adversarial inputs; the resulting vectors are committed under
// tests/golden/ (tests/golden/make_golden.py).
//
// What is restated (reference file:line -> here):
//   Object::Object                       segment.cc:5-21    -> init of lp[]/cls[]
//   AdjacencyRecord::AdjacencyRecord     segment.cc:24-46   -> make_record()
//   SortAndUpdateHash / Hasher           segment.cc:49-56, segment.h:237-242 -> rekey()
//   ComputeClassDeltaLogprob             segment.cc:107-142 -> rescore()
//   UpdateMergePriority                  segment.cc:145-150 -> rescore()
//   ObjectSegmenter::ObjectSegmenter     segment.cc:153-232 -> Segmenter::build()
//   RunSegmentation                      segment.cc:539-573 -> Segmenter::run()
//   Merge                                segment.cc:602-727 -> Segmenter::merge()
//   OutputMask                           segment.cc:491-517 -> Segmenter::emit()
//   ComputeTotalLogprobFromScratch       segment.cc:314-350 -> Segmenter::total_logprob()
//
// Layout differs from the reference on purpose (flat struct-of-arrays records, a parent
// forest instead of per-object pixel sets, no per-object heap allocation), but the three
// containers whose ORDER is observable are kept as the same standard containers fed the
// same operation sequence, so ties resolve exactly as in the reference:
//   * std::priority_queue keyed on the float priority only (heap mechanics decide ties),
//   * one std::unordered_map<size_t,int> per object keyed by id1*1619+id2*3203 (its
//     iteration order is the fold order inside a merge, hence the push order),
//   * one std::unordered_map<size_t,int> of live objects (its iteration order is the
//     label order of the output mask).
// Arithmetic is float32 exactly where the reference's is: logf for class and sameness
// terms, a double log for log(1-p) rounded to float, float sums, float priority.
// Build WITHOUT -ffast-math / -march flags that enable FMA contraction.

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <queue>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct ByPriority {
  bool operator()(const std::pair<float, int>& a, const std::pair<float, int>& b) const {
    return a.first < b.first;
  }
};

typedef std::unordered_map<size_t, int> AdjMap;

struct Segmenter {
  int C, O, W, H, N;
  float sdb, omf, bias;
  const float* cls_p;
  float* same_p;
  const int* offs;

  // objects (id = row*W + col)
  std::vector<float> lp;         // [N*C]
  std::vector<int> ocls;         // current class
  std::vector<int> osize;        // pixel count
  std::vector<float> osame;      // internal sameness log-prob
  std::vector<AdjMap> adj;       // per-object record map (order observable)
  std::vector<int> parent;       // absorbed -> survivor
  std::unordered_map<size_t, int> live;  // live objects (order observable)

  // records
  std::vector<int> r1, r2;       // endpoints, r1 < r2 by id; r2 = -1 once merged away
  std::vector<size_t> rkey;
  std::vector<float> rsame, rdiff, roml, rcdl, rprio;
  std::vector<int> rmcls;

  std::priority_queue<std::pair<float, int>, std::vector<std::pair<float, int> >, ByPriority> pq;

  long long n_pops = 0, n_merges = 0, n_rescored = 0, n_live_pops = 0;

  void rekey(int r) {
    if (r1[r] > r2[r]) std::swap(r1[r], r2[r]);
    rkey[r] = (size_t)r1[r] * 1619 + (size_t)r2[r] * 3203;
  }

  void rescore(int r) {
    const int a = r1[r], b = r2[r];
    if (ocls[a] == ocls[b]) {
      rcdl[r] = 0;
      rmcls[r] = ocls[a];
    } else {
      const float* la = &lp[(size_t)a * C];
      const float* lb = &lp[(size_t)b * C];
      int best = 0;
      float bestv = la[0] + lb[0];
      for (int c = 1; c < C; c++) {
        float v = la[c] + lb[c];
        if (v > bestv) { bestv = v; best = c; }
      }
      rmcls[r] = best;
      rcdl[r] = bestv - la[ocls[a]] - lb[ocls[b]];
    }
    size_t den = (size_t)osize[a] + (size_t)osize[b];
    rprio[r] = (roml[r] * omf + rcdl[r]) / den + bias;
    n_rescored++;
  }

  void build() {
    N = W * H;
    if (sdb != 0) {
      for (size_t i = 0; i < (size_t)O * N; i++) {
        float logit = logf(same_p[i]) - log(1.0 - same_p[i]) + sdb;
        same_p[i] = 1.0 / (1.0 + expf(-logit));
      }
    }
    lp.resize((size_t)N * C);
    ocls.resize(N);
    osize.assign(N, 1);
    osame.assign(N, 0.0f);
    adj.resize(N);
    parent.resize(N);
    for (int p = 0; p < N; p++) {
      float* l = &lp[(size_t)p * C];
      int best = 0;
      for (int c = 0; c < C; c++) {
        l[c] = logf(cls_p[(size_t)c * N + p]);
        if (l[c] > l[best]) best = c;
      }
      ocls[p] = best;
      parent[p] = p;
      live[(size_t)p] = p;
    }
    size_t cap = 0;
    for (int k = 0; k < O; k++) {
      long long hh = H - std::abs(offs[2 * k]), ww = W - std::abs(offs[2 * k + 1]);
      if (hh > 0 && ww > 0) cap += (size_t)(hh * ww);
    }
    r1.reserve(cap); r2.reserve(cap); rkey.reserve(cap); rsame.reserve(cap);
    rdiff.reserve(cap); roml.reserve(cap); rcdl.reserve(cap); rprio.reserve(cap);
    rmcls.reserve(cap);
    for (int row = 0; row < H; row++) {
      for (int col = 0; col < W; col++) {
        const int p = row * W + col;
        for (int k = 0; k < O; k++) {
          const int rr = row + offs[2 * k], cc = col + offs[2 * k + 1];
          if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
          const int q = rr * W + cc;
          const float sp = same_p[(size_t)k * N + p];
          const int r = (int)r1.size();
          r1.push_back(p); r2.push_back(q); rkey.push_back(0);
          rdiff.push_back((float)log(1.0 - sp));
          rsame.push_back(logf(sp));
          roml.push_back(rsame[r] - rdiff[r]);
          rcdl.push_back(0); rprio.push_back(0); rmcls.push_back(0);
          rekey(r);
          rescore(r);
          adj[p][rkey[r]] = r;
          adj[q][rkey[r]] = r;
          if (rprio[r] >= 0) pq.push(std::make_pair(rprio[r], r));
        }
      }
    }
  }

  void merge(int r) {
    int a = r1[r], b = r2[r];
    if (a < 0 || b < 0 || a == b) return;
    if (osize[a] < osize[b]) std::swap(a, b);   // a survives; tie keeps the lower id (r1)
    ocls[a] = rmcls[r];
    osize[a] += osize[b];
    float* la = &lp[(size_t)a * C];
    const float* lb = &lp[(size_t)b * C];
    for (int c = 0; c < C; c++) la[c] += lb[c];
    osame[a] += rsame[r] + osame[b];
    adj[a].erase(rkey[r]);
    adj[b].erase(rkey[r]);
    for (AdjMap::iterator it = adj[b].begin(); it != adj[b].end(); ++it) {
      const int t = it->second;
      int c3;
      if (r1[t] == b) { c3 = r2[t]; r1[t] = a; }
      else            { c3 = r1[t]; r2[t] = a; }
      const size_t old_key = rkey[t];
      rekey(t);
      adj[c3].erase(old_key);
      AdjMap::iterator hit = adj[a].find(rkey[t]);
      if (hit != adj[a].end()) {
        const int u = hit->second;
        roml[u] += roml[t];
        rdiff[u] += rdiff[t];
        rsame[u] += rsame[t];
        rprio[t] = std::numeric_limits<float>::min();   // tombstone
        rescore(u);
        if (rprio[u] >= 0) pq.push(std::make_pair(rprio[u], u));
      } else {
        adj[a][rkey[t]] = t;
        adj[c3][rkey[t]] = t;
        rescore(t);
        if (rprio[t] >= 0) pq.push(std::make_pair(rprio[t], t));
      }
    }
    AdjMap().swap(adj[b]);
    live.erase((size_t)b);
    parent[b] = a;
    r2[r] = -1;
    n_merges++;
  }

  void run() {
    while (!pq.empty()) {
      const float q = pq.top().first;
      const int r = pq.top().second;
      pq.pop();
      n_pops++;
      if (q != rprio[r]) continue;
      if (r2[r] < 0) continue;
      n_live_pops++;
      rescore(r);
      if (rprio[r] == q) merge(r);
      else if (rprio[r] >= 0) pq.push(std::make_pair(rprio[r], r));
    }
  }

  int root(int p) {
    int x = p;
    while (parent[x] != x) x = parent[x];
    while (parent[p] != x) { int nx = parent[p]; parent[p] = x; p = nx; }
    return x;
  }

  // label order = iteration order of the live-object map, as the reference's OutputMask
  void emit(int* output, int* object_class, int* partition) {
    std::vector<int> label(N, 0);
    for (int i = 0; i < N; i++) object_class[i] = -1;
    int k = 1;
    for (std::unordered_map<size_t, int>::iterator it = live.begin(); it != live.end(); ++it) {
      const int o = it->second;
      if (ocls[o] == 0) continue;
      object_class[k - 1] = ocls[o];
      label[o] = k++;
    }
    for (int p = 0; p < N; p++) {
      const int o = root(p);
      output[p] = label[o];
      if (partition) partition[p] = o;
    }
  }

  // float64 accumulation of the float32 terms on the final partition (before the
  // background collapse): sum_p lp[cls(obj(p))][p] + omf * (sum log p | log(1-p)).
  double total_logprob() {
    double cls_term = 0, same_term = 0, diff_term = 0;
    for (int p = 0; p < N; p++) {
      const int o = root(p);
      cls_term += (double)logf(cls_p[(size_t)ocls[o] * N + p]);
    }
    for (int row = 0; row < H; row++)
      for (int col = 0; col < W; col++) {
        const int p = row * W + col;
        const int op = root(p);
        for (int k = 0; k < O; k++) {
          const int rr = row + offs[2 * k], cc = col + offs[2 * k + 1];
          if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
          const float sp = same_p[(size_t)k * N + p];
          if (root(rr * W + cc) == op) same_term += (double)logf(sp);
          else diff_term += (double)(float)log(1.0 - sp);
        }
      }
    return cls_term + (diff_term + same_term) * (double)omf;
  }
};

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

// Extended entry: same inputs as the reference's c_run_segmentation (segment.cc:742-754;
// width precedes height; adj_pred is rewritten in place when same_different_bias != 0),
// plus optional outputs:
//   partition[H*W]  surviving object id (= lowest-history survivor pixel id) per pixel,
//                   BEFORE the class-0 collapse;
//   stats[10]       {total_logprob, n_objects, n_pops, n_merges, t_build_s, t_loop_s,
//                    n_rescored, n_initial_records, n_live_pops, 0}.
// Returns 0, or a negative code for invalid arguments (the reference would crash/exit).
int oracle_csegment_run(float* class_pred, int class_dim, float* adj_pred, int offset_dim,
                        int img_width, int img_height, int num_classes, const int* offset_list,
                        int* output, int* object_class, float same_different_bias,
                        float object_merge_factor, float merge_logprob_bias,
                        int* partition, double* stats) {
  if (!class_pred || !adj_pred || !offset_list || !output || !object_class) return -1;
  if (img_width <= 0 || img_height <= 0 || num_classes <= 0 || offset_dim < 0) return -2;
  if (class_dim < num_classes) return -3;
  if ((long long)img_width * img_height > (1LL << 30)) return -4;
  for (int a = 0; a < offset_dim; a++)
    for (int b = 0; b < offset_dim; b++) {
      const bool neg = offset_list[2 * a] == -offset_list[2 * b] &&
                       offset_list[2 * a + 1] == -offset_list[2 * b + 1];
      const bool dup = a != b && offset_list[2 * a] == offset_list[2 * b] &&
                       offset_list[2 * a + 1] == offset_list[2 * b + 1];
      if (neg || dup) return -5;   // o together with -o (or twice, or (0,0)): A.1 of SURVEY.md
    }
  Segmenter s;
  s.C = num_classes; s.O = offset_dim; s.W = img_width; s.H = img_height;
  s.sdb = same_different_bias; s.omf = object_merge_factor; s.bias = merge_logprob_bias;
  s.cls_p = class_pred; s.same_p = adj_pred; s.offs = offset_list;
  const double t0 = now_s();
  s.build();
  const size_t n_rec = s.r1.size();
  const double t1 = now_s();
  s.run();
  const double t2 = now_s();
  s.emit(output, object_class, partition);
  if (stats) {
    stats[0] = s.total_logprob();
    stats[1] = (double)s.live.size();
    stats[2] = (double)s.n_pops;
    stats[3] = (double)s.n_merges;
    stats[4] = t1 - t0;
    stats[5] = t2 - t1;
    stats[6] = (double)s.n_rescored;
    stats[7] = (double)n_rec;
    stats[8] = (double)s.n_live_pops;
    stats[9] = 0;
  }
  return 0;
}


// Phase A alone (the reference constructor, segment.cc:153-232): per-pixel arg-max class
// (Object::Object, :5-21), per in-bounds (pixel, offset) record its log-odds
// obj_merge_logprob (AdjacencyRecord ctor, :24-46) and its initial merge priority
// (ComputeClassDeltaLogprob + UpdateMergePriority, :107-150), in creation order (row-major
// pixels, offsets in list order, :209-231).  cls_out[N]; oml_out / prio_out [O][N] indexed by
// the SOURCE pixel, NaN where the edge leaves the image.  adj_pred is rewritten in place when
// same_different_bias != 0, as by the reference (:183-195).
int oracle_csegment_phase_a(float* class_pred, int class_dim, float* adj_pred, int offset_dim,
                            int img_width, int img_height, int num_classes,
                            const int* offset_list, float same_different_bias,
                            float object_merge_factor, float merge_logprob_bias, int* cls_out,
                            float* oml_out, float* prio_out) {
  if (!class_pred || !adj_pred || !offset_list || !cls_out || !oml_out || !prio_out) return -1;
  if (img_width <= 0 || img_height <= 0 || num_classes <= 0 || offset_dim < 0) return -2;
  if (class_dim < num_classes) return -3;
  const int C = num_classes, O = offset_dim, W = img_width, H = img_height, N = W * H;
  if (same_different_bias != 0) {
    for (size_t i = 0; i < (size_t)O * N; i++) {
      float logit = logf(adj_pred[i]) - log(1.0 - adj_pred[i]) + same_different_bias;
      adj_pred[i] = 1.0 / (1.0 + expf(-logit));
    }
  }
  std::vector<float> lp((size_t)N * C);
  for (int p = 0; p < N; p++) {
    float* l = &lp[(size_t)p * C];
    int best = 0;
    for (int c = 0; c < C; c++) {
      l[c] = logf(class_pred[(size_t)c * N + p]);
      if (l[c] > l[best]) best = c;
    }
    cls_out[p] = best;
  }
  const float nan = std::numeric_limits<float>::quiet_NaN();
  for (int k = 0; k < O; k++)
    for (int row = 0; row < H; row++)
      for (int col = 0; col < W; col++) {
        const int p = row * W + col;
        const int rr = row + offset_list[2 * k], cc = col + offset_list[2 * k + 1];
        const size_t e = (size_t)k * N + p;
        if (rr < 0 || rr >= H || cc < 0 || cc >= W) { oml_out[e] = nan; prio_out[e] = nan; continue; }
        const int q = rr * W + cc;
        const float sp = adj_pred[e];
        const float diff = (float)log(1.0 - sp);
        const float same = logf(sp);
        const float oml = same - diff;
        // record endpoints ordered by id (SortAndUpdateHash, segment.cc:49-56)
        const int a = p < q ? p : q, b = p < q ? q : p;
        float cdl = 0;
        if (cls_out[a] != cls_out[b]) {
          const float* la = &lp[(size_t)a * C];
          const float* lb = &lp[(size_t)b * C];
          float bestv = la[0] + lb[0];
          for (int c = 1; c < C; c++) { const float v = la[c] + lb[c]; if (v > bestv) bestv = v; }
          cdl = bestv - la[cls_out[a]] - lb[cls_out[b]];
        }
        const size_t den = 2;
        oml_out[e] = oml;
        prio_out[e] = (oml * object_merge_factor + cdl) / den + merge_logprob_bias;
      }
  return 0;
}

}  // extern "C"
