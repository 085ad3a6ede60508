// Decode/prefill projection GEMMs: a thin C-ABI over the
// hipBLASLt *library* kernels with a per-shape solution search.
//
// The reference leaves these to `F.linear` (P/layers/linear.py:32,103,124, embedding.py:98), i.e.
// to the BLAS library's default heuristic.  On gfx950 that heuristic is poor for the M <= 256
// weight-streaming shapes of a decode step (round-1 profile: 1.2-1.4 TB/s of weights,
// ~350 TFLOP/s at M = 256), so this file
//   * enumerates every library solution that supports the shape (hipblaslt_ext::getAllAlgos +
//     matmulIsAlgoSupported), times each on rotating weight buffers (so the 256 MiB Infinity
//     Cache cannot flatter a candidate: in a real step 28 GB stream between two uses of a
//     weight) and remembers the fastest per (M, N, K, ld*, dtype);
//   * launches the remembered solution (or the library's top heuristic for untuned shapes)
//     with a caller-provided workspace: no allocation, no sync => legal under stream capture.
// msgl_gemm_tune() itself synchronises and is an initialisation-time call.
//
// Layout: out[M, N] = x[M, K] . w[N, K]^T, all row-major (torch `F.linear`); in BLAS column-major
// terms D^T[N, M] = op_T(W)[N, K] . X^T[K, M]  => m = N, n = M, k = K, transA = T, transB = N.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/msgl_hip.h"

namespace {

thread_local char g_err[512] = "";

void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define GEMM_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) {              \
      set_err(__VA_ARGS__);     \
      return MSGL_EINVAL;       \
    }                           \
  } while (0)

#define GEMM_BLAS(call)                                                     \
  do {                                                                      \
    hipblasStatus_t st_ = (call);                                           \
    if (st_ != HIPBLAS_STATUS_SUCCESS) {                                    \
      set_err("%s failed with hipblasStatus %d (%s:%d)", #call, (int)st_, __FILE__, __LINE__); \
      return MSGL_ELAUNCH;                                                  \
    }                                                                       \
  } while (0)

#define GEMM_HIP(call)                                                      \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      set_err("%s failed: %s", #call, hipGetErrorString(e_));               \
      return MSGL_ELAUNCH;                                                  \
    }                                                                       \
  } while (0)

using Key = std::tuple<int, int, int, int, int64_t, int64_t, int64_t, int>;  // device, M N K, ldx ldw ldo, dtype

struct Problem {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, d = nullptr;
  ~Problem() {
    if (a) hipblasLtMatrixLayoutDestroy(a);
    if (b) hipblasLtMatrixLayoutDestroy(b);
    if (d) hipblasLtMatrixLayoutDestroy(d);
    if (desc) hipblasLtMatmulDescDestroy(desc);
  }
};

// hipblaslt_ext::Gemm carries data members; this file is compiled against the ROCm headers but may bind
// (same SONAME) to the libhipblaslt that PyTorch bundles.  The object is therefore constructed by the
// library's own exported constructor inside an oversized buffer and only ever touched through exported
// member functions, so a layout difference between the two versions cannot overrun it.
struct GemmBox {
  alignas(64) unsigned char raw[2048];
  bool live = false;  // a Gemm has been constructed in `raw`
  hipblaslt_ext::Gemm* get() { return reinterpret_cast<hipblaslt_ext::Gemm*>(raw); }
  ~GemmBox() {
    if (live) get()->~Gemm();  // the library's own exported destructor, like the constructor
  }
};

struct Plan {
  std::shared_ptr<Problem> prob;  // shared by the candidates of one search; freed with the last plan that refers to it
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  bool tuned = false;
  int algo_index = -1;
  int split_k = 0;          // 0: the solution's own setting (plain hipblasLtMatmul); > 0: ext Gemm + GemmTuning
  std::shared_ptr<GemmBox> box;   // split_k > 0 only; shared like `prob`
};

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;  // per device
// The tables live on the heap and are never destroyed: at process exit the library the plans' objects call into may
// already be gone.  msgl_gemm_reset_plans frees the objects of dropped plans explicitly.
std::map<Key, Plan>& g_plans = *new std::map<Key, Plan>();
// the best few candidates of the last search per shape, fastest first (msgl_gemm_finalists / msgl_gemm_select_finalist):
// back-to-back timing separates the top solutions by ~1 %, inside a captured decode step they differ by up to 9 %, so the
// host re-ranks them in place (engine.Engine.refine_plans_in_graph)
std::map<Key, std::vector<std::pair<float, Plan>>>& g_finalists = *new std::map<Key, std::vector<std::pair<float, Plan>>>();
// Untuned (heuristic) plans are created on first use of a shape; prefill M = total extend tokens takes almost any
// value, so a long-running server would grow the map without bound.  Tuned plans (decode shapes, a few dozen) are
// kept; untuned ones are dropped oldest-first beyond this many.
constexpr size_t kMaxUntunedPlans = 256;
std::deque<Key> g_untuned_order;

void release_plan(Plan& pl) {
  pl.prob.reset();
  pl.box.reset();
}

void remember_untuned(const Key& key) {
  g_untuned_order.push_back(key);
  while (g_untuned_order.size() > kMaxUntunedPlans) {
    const Key old = g_untuned_order.front();
    g_untuned_order.pop_front();
    auto it = g_plans.find(old);
    if (it != g_plans.end() && !it->second.tuned) {  // a shape tuned later keeps its plan
      release_plan(it->second);
      g_plans.erase(it);
    }
  }
}

int get_handle(hipblasLtHandle_t* h, int* dev_out) {
  int dev = 0;
  GEMM_HIP(hipGetDevice(&dev));
  auto it = g_handles.find(dev);
  if (it == g_handles.end()) {
    hipblasLtHandle_t nh;
    GEMM_BLAS(hipblasLtCreate(&nh));
    it = g_handles.emplace(dev, nh).first;
  }
  *h = it->second;
  *dev_out = dev;
  return MSGL_OK;
}

int make_problem(Problem* p, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype) {
  const hipDataType t = dtype == MSGL_BF16 ? HIP_R_16BF : HIP_R_16F;
  GEMM_BLAS(hipblasLtMatmulDescCreate(&p->desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  const int32_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  GEMM_BLAS(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
  GEMM_BLAS(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
  // A = W stored [N][K] row-major = column-major K x N with ld = ldw (used transposed)
  GEMM_BLAS(hipblasLtMatrixLayoutCreate(&p->a, t, (uint64_t)K, (uint64_t)N, ldw));
  // B = X stored [M][K] row-major = column-major K x M with ld = ldx
  GEMM_BLAS(hipblasLtMatrixLayoutCreate(&p->b, t, (uint64_t)K, (uint64_t)M, ldx));
  // D = out stored [M][N] row-major = column-major N x M with ld = ldo
  GEMM_BLAS(hipblasLtMatrixLayoutCreate(&p->d, t, (uint64_t)N, (uint64_t)M, ldo));
  return MSGL_OK;
}

int check_args(const void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
               int64_t ldo, int dtype) {
  GEMM_REQUIRE(out && x && w, "gemm: null pointer");
  GEMM_REQUIRE(M >= 1 && N >= 1 && K >= 1, "gemm: bad shape %d x %d x %d", M, N, K);
  GEMM_REQUIRE(ldx >= K && ldw >= K && ldo >= N, "gemm: leading dimensions (%lld, %lld, %lld) too small",
               (long long)ldx, (long long)ldw, (long long)ldo);
  GEMM_REQUIRE(dtype == MSGL_BF16 || dtype == MSGL_FP16, "gemm: unsupported dtype code %d", dtype);
  return MSGL_OK;
}

int run(hipblasLtHandle_t h, const Plan& pl, void* out, const void* x, const void* w, void* ws, size_t ws_bytes,
        hipStream_t s) {
  const float alpha = 1.0f, beta = 0.0f;
  if (pl.workspace > ws_bytes) {
    set_err("gemm: solution needs %zu workspace bytes, caller gave %zu", pl.workspace, ws_bytes);
    return MSGL_EINVAL;
  }
  if (pl.split_k > 0) {
    // split-K over workgroups (Tensile "GSU"): host-side argument setup, then plain kernel launches
    hipblaslt_ext::Gemm* g = pl.box->get();
    GEMM_BLAS(g->setProblem(pl.prob->desc, &alpha, w, pl.prob->a, x, pl.prob->b, &beta, out, pl.prob->d, out,
                            pl.prob->d));
    hipblaslt_ext::GemmTuning tuning;
    tuning.setSplitK((uint16_t)pl.split_k);
    GEMM_BLAS(g->initialize(pl.algo, tuning, ws, true, s));
    GEMM_BLAS(g->run(s));
    return MSGL_OK;
  }
  GEMM_BLAS(hipblasLtMatmul(h, pl.prob->desc, &alpha, w, pl.prob->a, x, pl.prob->b, &beta, out, pl.prob->d, out,
                            pl.prob->d, &pl.algo, ws, ws_bytes, s));
  return MSGL_OK;
}

// heuristic top-1 under the workspace limit
int heuristic_plan(hipblasLtHandle_t h, Plan* pl, size_t ws_bytes) {
  hipblasLtMatmulPreference_t pref;
  GEMM_BLAS(hipblasLtMatmulPreferenceCreate(&pref));
  uint64_t lim = ws_bytes;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &lim, sizeof(lim));
  hipblasLtMatmulHeuristicResult_t res[1];
  int n = 0;
  hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(h, pl->prob->desc, pl->prob->a, pl->prob->b, pl->prob->d,
                                                       pl->prob->d, pref, 1, res, &n);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || n < 1) {
    set_err("gemm: the library has no solution for this shape (status %d)", (int)st);
    return MSGL_EINVAL;
  }
  pl->algo = res[0].algo;
  pl->workspace = res[0].workspaceSize;
  pl->algo_index = hipblaslt_ext::getIndexFromAlgo(pl->algo);
  return MSGL_OK;
}

}  // namespace

extern "C" {

const char* msgl_gemm_last_error(void) { return g_err; }

int msgl_gemm_nt(void* out, const void* x, const void* w, int M, int N, int K, int64_t ldx, int64_t ldw,
                 int64_t ldo, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  int rc = check_args(out, x, w, M, N, K, ldx, ldw, ldo, dtype);
  if (rc != MSGL_OK) return rc;
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t h;
  int dev;
  if ((rc = get_handle(&h, &dev)) != MSGL_OK) return rc;
  const size_t ws_bytes = workspace ? (size_t)std::max<int64_t>(workspace_bytes, 0) : 0;
  const Key key{dev, M, N, K, ldx, ldw, ldo, dtype};
  auto it = g_plans.find(key);
  if (it == g_plans.end()) {
    Plan pl;
    pl.prob = std::make_shared<Problem>();
    if ((rc = make_problem(pl.prob.get(), M, N, K, ldx, ldw, ldo, dtype)) != MSGL_OK) return rc;
    if ((rc = heuristic_plan(h, &pl, ws_bytes)) != MSGL_OK) return rc;
    it = g_plans.emplace(key, pl).first;
    remember_untuned(key);
    it = g_plans.find(key);  // (the bound never evicts the entry just added: it is the newest)
  }
  return run(h, it->second, out, x, w, workspace, ws_bytes, static_cast<hipStream_t>(stream));
}

int msgl_gemm_tune(void* out, const void* x, const void* const* w_list, int n_w, int M, int N, int K, int64_t ldx,
                   int64_t ldw, int64_t ldo, int dtype, void* workspace, int64_t workspace_bytes,
                   int max_candidates, int split_k_search, int iters, float* best_us, float* default_us,
                   int* best_index, int* best_split_k, int* n_tried, void* stream) {
  GEMM_REQUIRE(w_list && n_w >= 1, "gemm_tune: need at least one weight buffer");
  int rc = check_args(out, x, w_list[0], M, N, K, ldx, ldw, ldo, dtype);
  if (rc != MSGL_OK) return rc;
  if (iters < 1) iters = 1;
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t h;
  int dev;
  if ((rc = get_handle(&h, &dev)) != MSGL_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t ws_bytes = workspace ? (size_t)std::max<int64_t>(workspace_bytes, 0) : 0;

  Plan base;
  base.prob = std::make_shared<Problem>();
  if ((rc = make_problem(base.prob.get(), M, N, K, ldx, ldw, ldo, dtype)) != MSGL_OK) return rc;
  if ((rc = heuristic_plan(h, &base, ws_bytes)) != MSGL_OK) return rc;

  const hipDataType t = dtype == MSGL_BF16 ? HIP_R_16BF : HIP_R_16F;
  // max_candidates: 0 = every library solution, n > 1 = the first n of them, -n = the top n of the
  // library's own heuristic ranking (cheap: tens of candidates), 1 = heuristic pick only
  std::vector<hipblasLtMatmulHeuristicResult_t> all;
  // split-K variants are tried in every search mode: the heuristic ranking never proposes them, and they are
  // what the K = 17408 down-projection needs at every M (77 vs 127 us at M = 64 .. 256)
  const bool split_search = max_candidates != 1 && split_k_search != 0;
  if (max_candidates == 0 || max_candidates > 1) {
    hipblasStatus_t st = hipblaslt_ext::getAllAlgos(h, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T,
                                                    HIPBLAS_OP_N, t, t, t, t, HIPBLAS_COMPUTE_32F, all);
    if (st != HIPBLAS_STATUS_SUCCESS) all.clear();
  } else if (max_candidates < 0) {
    hipblasLtMatmulPreference_t pref;
    GEMM_BLAS(hipblasLtMatmulPreferenceCreate(&pref));
    uint64_t lim = ws_bytes;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &lim, sizeof(lim));
    all.resize((size_t)(-max_candidates));
    int n = 0;
    hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(h, base.prob->desc, base.prob->a, base.prob->b,
                                                         base.prob->d, base.prob->d, pref, -max_candidates,
                                                         all.data(), &n);
    hipblasLtMatmulPreferenceDestroy(pref);
    all.resize(st == HIPBLAS_STATUS_SUCCESS ? (size_t)std::max(n, 0) : 0);
    max_candidates = 0;
  }
  // candidate 0 is always the library's own heuristic pick (= what F.linear would run)
  std::vector<Plan> cands;
  cands.push_back(base);
  const float alpha = 1.0f, beta = 0.0f;
  for (auto& r : all) {
    if (max_candidates > 0 && (int)cands.size() >= max_candidates) break;
    size_t need = 0;
    if (hipblaslt_ext::matmulIsAlgoSupported(h, base.prob->desc, &alpha, base.prob->a, base.prob->b, &beta,
                                             base.prob->d, base.prob->d, r.algo, need) != HIPBLAS_STATUS_SUCCESS)
      continue;
    if (need > ws_bytes) continue;
    Plan c = base;
    c.algo = r.algo;
    c.workspace = need;
    c.algo_index = hipblaslt_ext::getIndexFromAlgo(c.algo);
    if (c.algo_index == base.algo_index) continue;
    cands.push_back(c);
  }

  // split-K variants of every supported solution (the few-tile shapes N <= 8192 of a decode step leave
  // most CUs idle otherwise).
  if (split_search) {
    auto box = std::make_shared<GemmBox>();
    new (box->raw) hipblaslt_ext::Gemm(h, base.prob->desc, &alpha, w_list[0], base.prob->a, x, base.prob->b, &beta,
                                       out, base.prob->d, out, base.prob->d);
    box->live = true;
    const size_t n_plain = cands.size();
    static const int kSplits[] = {2, 3, 4, 6, 8, 12, 16};
    for (size_t i = 0; i < n_plain; ++i) {
      for (int sk : kSplits) {
        if (K / sk < 512) break;
        hipblaslt_ext::GemmTuning tuning;
        tuning.setSplitK((uint16_t)sk);
        size_t need = 0;
        hipblasLtMatmulAlgo_t algo = cands[i].algo;
        if (box->get()->isAlgoSupported(algo, tuning, need) != HIPBLAS_STATUS_SUCCESS) continue;
        if (need > ws_bytes) continue;
        Plan c = cands[i];
        c.algo = algo;
        c.workspace = need;
        c.split_k = sk;
        c.box = box;
        cands.push_back(c);
      }
    }
  }

  hipEvent_t e0, e1;
  GEMM_HIP(hipEventCreate(&e0));
  GEMM_HIP(hipEventCreate(&e1));
  // Each repetition is bracketed by its own event pair: the split-K path does host-side argument setup
  // per call, which must not be billed to the kernel (under hipGraph replay it does not exist).
  auto time_us = [&](const Plan& c, int reps, float* us) -> int {
    int r0 = run(h, c, out, x, w_list[0], workspace, ws_bytes, s);  // warm-up (code object load)
    if (r0 != MSGL_OK) return r0;
    float total_ms = 0.f;
    for (int i = 0; i < reps; ++i) {
      if (hipEventRecord(e0, s) != hipSuccess) return MSGL_ELAUNCH;
      r0 = run(h, c, out, x, w_list[(i + 1) % n_w], workspace, ws_bytes, s);
      if (r0 != MSGL_OK) return r0;
      if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return MSGL_ELAUNCH;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return MSGL_ELAUNCH;
      total_ms += ms;
    }
    *us = total_ms * 1e3f / reps;
    return MSGL_OK;
  };

  // pass 1: a short timing of every candidate; pass 2: re-time the best few with full iters
  std::vector<std::pair<float, int>> ranked;
  const int quick = std::max(2, std::min(iters, 4));
  for (int i = 0; i < (int)cands.size(); ++i) {
    float us = 0.f;
    if (time_us(cands[i], quick, &us) != MSGL_OK) {
      (void)hipGetLastError();
      continue;
    }
    ranked.emplace_back(us, i);
  }
  if (ranked.empty()) {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    set_err("gemm_tune: no candidate ran");
    return MSGL_ELAUNCH;
  }
  std::sort(ranked.begin(), ranked.end());
  // finals: three interleaved rounds over the best dozen, per-candidate minimum (run-to-run noise of a
  // single timing is a few percent, more than the spread between the top candidates)
  float best = 1e30f, def = -1.f;
  int best_i = 0;
  const int finals = std::min<int>(12, ranked.size());
  std::vector<float> fin((size_t)finals, 1e30f);
  bool base_in_finals = false;
  for (int round = 0; round < 3; ++round) {
    for (int r = 0; r < finals; ++r) {
      float us = 0.f;
      if (time_us(cands[ranked[r].second], iters, &us) != MSGL_OK) continue;
      fin[r] = std::min(fin[r], us);
    }
  }
  for (int r = 0; r < finals; ++r) {
    if (ranked[r].second == 0) {
      def = fin[r];
      base_in_finals = true;
    }
    if (fin[r] < best) {
      best = fin[r];
      best_i = ranked[r].second;
    }
  }
  if (!base_in_finals) {
    for (int round = 0; round < 3; ++round) {
      float us = 0.f;
      if (time_us(cands[0], iters, &us) == MSGL_OK) def = def < 0.f ? us : std::min(def, us);
    }
    if (def >= 0.f && def < best) {
      best = def;
      best_i = 0;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);

  Plan chosen = cands[best_i];
  chosen.tuned = true;
  const Key key{dev, M, N, K, ldx, ldw, ldo, dtype};
  g_plans[key] = chosen;
  {
    std::vector<std::pair<float, Plan>> fl;
    for (int r = 0; r < finals; ++r)
      if (fin[r] < 1e29f) {
        Plan c = cands[ranked[r].second];
        c.tuned = true;
        fl.emplace_back(fin[r], c);
      }
    std::sort(fl.begin(), fl.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    if (fl.size() > 6) fl.resize(6);
    g_finalists[key] = fl;
  }
  if (best_us) *best_us = best;
  if (default_us) *default_us = def;
  if (best_index) *best_index = chosen.algo_index;
  if (best_split_k) *best_split_k = chosen.split_k;
  if (n_tried) *n_tried = (int)ranked.size();
  return MSGL_OK;
}

// Forget every plan (tuned or not): the next call of a shape takes the library's heuristic again.  Plans are process
// state; an engine that asks for gemm_tune = "off" must not inherit what an earlier engine searched.
int msgl_gemm_reset_plans(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  // the problem descriptors / split-K objects are reference-counted: dropped with the last plan or finalist holding them
  g_plans.clear();
  g_finalists.clear();
  g_untuned_order.clear();
  return MSGL_OK;
}

// The best candidates of the last msgl_gemm_tune of this shape, fastest first: writes up to max_n times (us) and returns
// how many there are (0 if the shape was never searched).
int msgl_gemm_finalists(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, float* us_out, int max_n) {
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t h;
  int dev;
  int rc = get_handle(&h, &dev);
  if (rc != MSGL_OK) return rc;
  auto it = g_finalists.find(Key{dev, M, N, K, ldx, ldw, ldo, dtype});
  if (it == g_finalists.end()) return 0;
  const int n = (int)it->second.size();
  for (int i = 0; i < n && i < max_n && us_out; ++i) us_out[i] = it->second[(size_t)i].first;
  return n;
}

// Make finalist `index` the shape's plan (what msgl_gemm_nt launches from now on).
int msgl_gemm_select_finalist(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, int index) {
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t h;
  int dev;
  int rc = get_handle(&h, &dev);
  if (rc != MSGL_OK) return rc;
  const Key key{dev, M, N, K, ldx, ldw, ldo, dtype};
  auto it = g_finalists.find(key);
  GEMM_REQUIRE(it != g_finalists.end() && index >= 0 && index < (int)it->second.size(),
               "gemm_select_finalist: shape has no finalist %d", index);
  g_plans[key] = it->second[(size_t)index].second;
  return MSGL_OK;
}

// The shape's current plan as (library solution index, split-K factor): returns 1 and fills the two if the shape has a
// SEARCHED plan, 0 if it has none (heuristic / never seen).  With msgl_gemm_set_plan this carries one process's search result
// into another process (the reference-driven parity tests replay a forward on exactly the plans its recorder ran).
int msgl_gemm_get_plan(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, int* algo_index, int* split_k) {
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t h;
  int dev;
  int rc = get_handle(&h, &dev);
  if (rc != MSGL_OK) return rc;
  auto it = g_plans.find(Key{dev, M, N, K, ldx, ldw, ldo, dtype});
  if (it == g_plans.end() || !it->second.tuned) return 0;
  if (algo_index) *algo_index = it->second.algo_index;
  if (split_k) *split_k = it->second.split_k;
  return 1;
}

// Make library solution `algo_index` (with split-K factor `split_k`, 0 = the solution's own) the shape's plan without a
// search.  Fails if this library build does not know the index or the solution does not support the shape within
// `workspace_bytes` of scratch.
int msgl_gemm_set_plan(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, int algo_index, int split_k,
                       void* workspace, int64_t workspace_bytes) {
  GEMM_REQUIRE(M >= 1 && N >= 1 && K >= 1 && (dtype == MSGL_BF16 || dtype == MSGL_FP16) && algo_index >= 0 && split_k >= 0,
               "gemm_set_plan: bad arguments");
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t h;
  int dev;
  int rc = get_handle(&h, &dev);
  if (rc != MSGL_OK) return rc;
  Plan pl;
  pl.prob = std::make_shared<Problem>();
  if ((rc = make_problem(pl.prob.get(), M, N, K, ldx, ldw, ldo, dtype)) != MSGL_OK) return rc;
  std::vector<int> idx{algo_index};
  std::vector<hipblasLtMatmulHeuristicResult_t> res;
  GEMM_BLAS(hipblaslt_ext::getAlgosFromIndex(h, idx, res));
  GEMM_REQUIRE(!res.empty(), "gemm_set_plan: the library has no solution with index %d", algo_index);
  const float alpha = 1.0f, beta = 0.0f;
  const size_t ws_bytes = (size_t)std::max<int64_t>(workspace_bytes, 0);
  size_t need = 0;
  pl.algo = res[0].algo;
  if (split_k > 0) {
    // the split-K object wants operand pointers at construction (the library refuses null ones); run() sets the real ones
    // before every launch, so the caller's scratch buffer stands in for all three here
    GEMM_REQUIRE(workspace != nullptr, "gemm_set_plan: a split-K plan needs the workspace pointer");
    auto box = std::make_shared<GemmBox>();
    new (box->raw) hipblaslt_ext::Gemm(h, pl.prob->desc, &alpha, workspace, pl.prob->a, workspace, pl.prob->b, &beta, workspace,
                                       pl.prob->d, workspace, pl.prob->d);
    box->live = true;
    hipblaslt_ext::GemmTuning tuning;
    tuning.setSplitK((uint16_t)split_k);
    GEMM_REQUIRE(box->get()->isAlgoSupported(pl.algo, tuning, need) == HIPBLAS_STATUS_SUCCESS,
                 "gemm_set_plan: solution %d does not support the shape with split-K %d", algo_index, split_k);
    pl.box = box;
  } else {
    GEMM_REQUIRE(hipblaslt_ext::matmulIsAlgoSupported(h, pl.prob->desc, &alpha, pl.prob->a, pl.prob->b, &beta, pl.prob->d,
                                                      pl.prob->d, pl.algo, need) == HIPBLAS_STATUS_SUCCESS,
                 "gemm_set_plan: solution %d does not support the shape", algo_index);
  }
  GEMM_REQUIRE(need <= ws_bytes, "gemm_set_plan: solution %d needs %zu workspace bytes, %zu given", algo_index, need, ws_bytes);
  pl.workspace = need;
  pl.algo_index = algo_index;
  pl.split_k = split_k;
  pl.tuned = true;
  const Key key{dev, M, N, K, ldx, ldw, ldo, dtype};
  g_plans[key] = pl;
  g_finalists.erase(key);  // an imported plan has no search behind it: a later select_finalist(0) must not silently replace it
  return MSGL_OK;
}

int msgl_gemm_solution_name(int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int dtype, char* buf,
                            int buf_len) {
  GEMM_REQUIRE(buf && buf_len > 0, "gemm_solution_name: bad buffer");
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t h;
  int dev;
  int rc = get_handle(&h, &dev);
  if (rc != MSGL_OK) return rc;
  auto it = g_plans.find(Key{dev, M, N, K, ldx, ldw, ldo, dtype});
  GEMM_REQUIRE(it != g_plans.end(), "gemm_solution_name: shape not planned yet");
  std::string name = hipblaslt_ext::getKernelNameFromAlgo(h, it->second.algo);
  snprintf(buf, (size_t)buf_len, "%s[splitK %d] %s", it->second.tuned ? "[tuned]" : "[heuristic]", it->second.split_k,
           name.c_str());
  return it->second.algo_index;
}

}  // extern "C"
