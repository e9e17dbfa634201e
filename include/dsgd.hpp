// C++ host-side mirror of the reference's surface for the hot path, over the C ABI of dsgd.h (header only, C++17).
//
// The reference is Scala (compiled JVM code) and this image has no JVM, so the host side above the C ABI is offered in
// C++ with the reference's names, argument meaning and error behaviour -- the code a JNI-free native host would write,
// and what tests/cpp/host_mirror_test.cpp drives so that the parity tests read like the reference's own.
// Citations are relative to /root/reference/src/main/scala/epfl/distributed/.
//
//   Vec                      dense float[D+1] indexed by key (the wire type Sparse, proto.proto:28-31, at the ABI)
//   SparseSVM                core/ml/SparseSVM.scala:11-31 with the data resident on the device
//   Slave                    core/Slave.scala:113-198 (handlers of SlaveImpl)
//   Master::fit              core/Master.scala:120-218
//   SplitStrategy::vanilla   core/ml/SplitStrategy.scala:13-14
//   EarlyStopping            core/ml/EarlyStopping.scala:11-46
//   GradState                core/ml/GradState.scala:6-24
//   JavaRandom, shuffle      java.util.Random / scala.util.Random.shuffle (2.12), seeded 0 at Main.scala:32
#ifndef DSGD_HPP
#define DSGD_HPP
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dsgd.h"

namespace dsgd {

using Vec = std::vector<float>;

// what `require` throws (math/Vec.scala:129, math/Sparse.scala:16) / Sparse.apply on a bad key (math/Sparse.scala:63)
struct IllegalArgumentException : std::invalid_argument {
  using std::invalid_argument::invalid_argument;
};
struct IndexOutOfBoundsException : std::out_of_range {
  using std::out_of_range::out_of_range;
};
struct NativeError : std::runtime_error {
  int code;
  NativeError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
  if (rc == DSGD_OK) return;
  const std::string msg = dsgd_last_error();
  if (rc == DSGD_EINVAL) throw IllegalArgumentException(msg);
  if (rc == DSGD_ERANGE) throw IndexOutOfBoundsException(msg);
  throw NativeError(rc, msg);
}

// ---- java.util.Random (48-bit LCG) and scala.util.Random.shuffle ------------------------------------------------------
class JavaRandom {
 public:
  explicit JavaRandom(int64_t seed = 0) : seed_((seed ^ 0x5DEECE66DLL) & ((1LL << 48) - 1)) {}
  int32_t next(int bits) {
    seed_ = (seed_ * 0x5DEECE66DLL + 0xBLL) & ((1LL << 48) - 1);
    return (int32_t)(seed_ >> (48 - bits));
  }
  int32_t nextInt() { return next(32); }
  int32_t nextInt(int32_t bound) {
    if (bound <= 0) throw IllegalArgumentException("bound must be positive");
    int32_t r = next(31);
    const int32_t m = bound - 1;
    if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
    for (int32_t u = r; u - (r = u % bound) + m < 0; u = next(31)) {
    }
    return r;
  }
  // the 48-bit internal state (what dsgd_plan_create_from_seed takes and hands back: the device draws an epoch's lists)
  uint64_t state() const { return (uint64_t)seed_; }
  void setState(uint64_t s) { seed_ = (int64_t)(s & ((1ULL << 48) - 1)); }

 private:
  int64_t seed_;
};
// scala.util.Random.shuffle (2.12): copy, then for n = len .. 2: swap(n - 1, nextInt(n))
template <class T>
std::vector<T> shuffle(std::vector<T> xs, JavaRandom& rnd) {
  for (int32_t n = (int32_t)xs.size(); n >= 2; --n) std::swap(xs[(size_t)n - 1], xs[(size_t)rnd.nextInt(n)]);
  return xs;
}

// ---- core/ml/SplitStrategy.scala:13-14 ---------------------------------------------------------------------------------
namespace SplitStrategy {
// indices.grouped(ceil(n / nSlaves)): contiguous ranges, the last may be shorter, there may be FEWER than nSlaves
inline std::vector<std::pair<int64_t, int64_t>> vanilla(int64_t n, int nSlaves) {
  std::vector<std::pair<int64_t, int64_t>> out;
  const int64_t size = (n + nSlaves - 1) / nSlaves;
  for (int64_t b = 0; b < n; b += size) out.emplace_back(b, std::min(n, b + size));
  return out;
}
}  // namespace SplitStrategy

// ---- core/ml/EarlyStopping.scala:11-46 (losses NEWEST FIRST) -------------------------------------------------------------
namespace EarlyStopping {
using Criterion = std::function<bool(const std::deque<double>&)>;
inline Criterion target(double t) {
  return [t](const std::deque<double>& losses) { return !losses.empty() && losses.front() <= t; };
}
inline Criterion noImprovement(int patience = 5, double minDelta = 1e-3, std::optional<int> minSteps = std::nullopt) {
  return [=](const std::deque<double>& losses) {
    const double absMinDelta = std::fabs(minDelta);
    auto checkNoImprovement = [&]() {
      double mn = std::numeric_limits<double>::max();
      long idxMin = -1;
      long index = 0;
      for (double num : losses) {
        if (num - mn <= absMinDelta) {
          mn = num;
          idxMin = index;
        }
        ++index;
      }
      return idxMin == 0 ? false : idxMin >= patience;
    };
    if (losses.empty()) return false;
    if (!minSteps) return checkNoImprovement();
    return *minSteps < (int)losses.size() ? false : checkNoImprovement();
  };
}
}  // namespace EarlyStopping

// ---- core/ml/GradState.scala:6-24 (`grad` holds the WEIGHTS) -----------------------------------------------------------
struct GradState {
  Vec grad;
  int64_t updates = 0;
  std::optional<double> loss;
  bool finished = false;
  static GradState start(Vec w) { return GradState{std::move(w), 0, std::nullopt, false}; }
  GradState replaceGrad(Vec w) const { return GradState{std::move(w), updates + 1, loss, finished}; }
  GradState finish(std::optional<double> l) const { return GradState{grad, updates, l, true}; }
};

// ---- the data a node holds: Array[(Vec, Int)] (utils/Dataset.scala:11) in CSR form, 1-based feature ids as keys -------
struct Data {
  int64_t n_rows = 0;
  std::vector<int64_t> row_ptr{0};
  std::vector<int32_t> col;
  std::vector<float> val;
  std::vector<int8_t> label;
  void add(const std::vector<std::pair<int32_t, float>>& x, int y) {
    for (const auto& kv : x) {
      col.push_back(kv.first);
      val.push_back(kv.second);
    }
    row_ptr.push_back((int64_t)col.size());
    label.push_back((int8_t)y);
    ++n_rows;
  }
};

// ---- core/ml/SparseSVM.scala:11-31 -----------------------------------------------------------------------------------
// class SparseSVM(lambda, dimSparsity): forward / loss / backward / regularize over samples of the resident data.
class SparseSVM {
 public:
  SparseSVM(double lambda, int nFeatures, int device = 0, unsigned flags = 0) : lambda_(lambda), d_(nFeatures) {
    dsgd_config cfg{};
    cfg.n_features = nFeatures;
    cfg.device = device;
    cfg.lambda = lambda;
    cfg.flags = flags;
    check(dsgd_create(&cfg, &ctx_));
  }
  SparseSVM(const SparseSVM&) = delete;
  SparseSVM& operator=(const SparseSVM&) = delete;
  ~SparseSVM() { dsgd_destroy(ctx_); }

  int size() const { return d_; }
  double lambda() const { return lambda_; }
  dsgd_ctx* ctx() const { return ctx_; }

  void load(const Data& data) {
    check(dsgd_load_csr(ctx_, data.n_rows, data.row_ptr.data(), data.col.data(), data.val.data(), data.label.data()));
    nRows_ = data.n_rows;
  }
  // Main.scala:54-65: dimSparsity from the first nTrain rows (incl. its off-by-one)
  Vec buildDimSparsity(int64_t nTrain) {
    Vec ds((size_t)d_ + 1);
    check(dsgd_build_dim_sparsity(ctx_, nTrain, ds.data()));
    return ds;
  }
  void setDimSparsity(const Vec& ds) {
    requireSize(ds);
    check(dsgd_set_dim_sparsity(ctx_, ds.data()));
  }
  // forward(w, x) for every sample: -signum(x . w)   (:14)
  std::vector<float> forward(const Vec& w, const std::vector<int32_t>& samplesIdx) {
    requireSize(w);
    std::vector<float> pred(samplesIdx.size());
    check(dsgd_forward(ctx_, w.data(), samplesIdx.data(), (int64_t)samplesIdx.size(), pred.data()));
    return pred;
  }
  // regularize(Vec.sum(samples.map(backward(w, _))), w)   (:26-31 + core/Slave.scala:147-155)
  Vec gradient(const Vec& w, const std::vector<int32_t>& samplesIdx, dsgd_batch_stats* stats = nullptr) {
    requireSize(w);
    if (samplesIdx.empty()) throw IllegalArgumentException("requirement failed");  // Vec.sum of nothing (math/Vec.scala:129)
    Vec g((size_t)d_ + 1);
    check(dsgd_gradient(ctx_, w.data(), samplesIdx.data(), (int64_t)samplesIdx.size(), g.data(), stats));
    return g;
  }
  // loss(w, samples) = lambda |w|^2 + mean hinge of the predictions; accuracy = mean [pred == y]   (:16-23)
  double loss(const Vec& w, int64_t rowBegin, int64_t rowEnd) {
    requireSize(w);
    double l = 0, a = 0;
    check(dsgd_loss_acc(ctx_, w.data(), rowBegin, rowEnd, &l, &a, nullptr));
    return l;
  }
  double accuracy(const Vec& w, int64_t rowBegin, int64_t rowEnd) {
    requireSize(w);
    double l = 0, a = 0;
    check(dsgd_loss_acc(ctx_, w.data(), rowBegin, rowEnd, &l, &a, nullptr));
    return a;
  }

 private:
  void requireSize(const Vec& v) const {
    if ((int)v.size() != d_ + 1) throw IllegalArgumentException("requirement failed: Can't perform operation: vectors have different sizes");
  }
  double lambda_;
  int d_;
  int64_t nRows_ = 0;
  dsgd_ctx* ctx_ = nullptr;
};

// ---- core/Slave.scala:113-198: the request / reply types of proto.proto and the handlers ---------------------------------
struct GradientRequest {
  Vec weights;
  std::vector<int32_t> samples;
};
struct ForwardRequest {
  std::vector<int32_t> samples;
  Vec weights;
};
struct StartAsyncRequest {
  Vec weights;
  std::vector<std::pair<int64_t, int64_t>> samples;  // contiguous ranges of assigned rows (SplitStrategy.vanilla)
  int batchSize = 1;
  double learningRate = 0;
};
struct GradUpdate {
  Vec gradUpdate;
};
struct ForwardReply {
  std::vector<float> predictions;
};

class Slave {
 public:
  Slave(SparseSVM& model, bool async) : model_(model), async_(async) {}

  GradUpdate gradient(const GradientRequest& request) { return GradUpdate{model_.gradient(request.weights, request.samples)}; }  // :142
  ForwardReply forward(const ForwardRequest& request) { return ForwardReply{model_.forward(request.weights, request.samples)}; }  // :129

  void startAsync(const StartAsyncRequest& request, int64_t maxUpdates, uint64_t seed = 0) {  // :159-175
    if (!async_) throw IllegalArgumentException("requirement failed: Cannot initialize async computation: slave is in synchronous mode.");
    if (runningAsync()) throw IllegalArgumentException("requirement failed: Async computation already running, can't be initialized unless stopped first");
    check(dsgd_set_weights(model_.ctx(), request.weights.data()));
    std::vector<int64_t> b, e;
    for (const auto& r : request.samples) {
      b.push_back(r.first);
      e.push_back(r.second);
    }
    check(dsgd_async_start(model_.ctx(), b.data(), e.data(), (int32_t)b.size(), request.batchSize, (float)request.learningRate,
                           maxUpdates, seed, /*positional_bug=*/1));
  }
  void updateGrad(const std::vector<int32_t>& keys, const std::vector<float>& values) {  // :177-185: weights - gradUpdate
    if (!async_) throw IllegalArgumentException("requirement failed: Cannot update gradient: slave is in synchronous mode.");
    if (keys.size() != values.size()) throw IllegalArgumentException("requirement failed");
    check(dsgd_update_grad(model_.ctx(), keys.data(), values.data(), (int64_t)keys.size()));
  }
  void stopAsync() {  // :187-196
    if (!async_) throw IllegalArgumentException("requirement failed: Cannot stop async computation: slave is in synchronous mode.");
    check(dsgd_async_stop(model_.ctx()));
  }
  bool runningAsync() {
    int64_t u = 0;
    int32_t running = 0;
    check(dsgd_async_updates(model_.ctx(), &u, &running));
    return running != 0;
  }

 private:
  SparseSVM& model_;
  bool async_;
};

// ---- core/Master.scala:120-218 for the workers hosted behind ONE context ------------------------------------------------
// rows [0, nTrain) are the train set, [nTrain, nRows) the test set (Main.scala:52)
class Master {
 public:
  Master(SparseSVM& model, int64_t nTrain, int64_t nRows, int nodeCount, JavaRandom rnd = JavaRandom(0))
      : model_(model), nTrain_(nTrain), nRows_(nRows), nodeCount_(nodeCount), rnd_(rnd) {}

  std::deque<double> losses, accs, testLosses, testAccs;  // newest first
  bool usePlans = true;   // an epoch's batches as ONE resident plan (dsgd_plan_create / dsgd_plan_run); false: one dsgd_sync_step per batch
  // ... and the plan's lists DRAWN BY THE DEVICE, draw for draw the same stream (dsgd_plan_create_from_seed), for epochs of at
  // least this many draws (the device form has ~1.5 ms of fixed cost; < 0: never); outside its limits the host draws as ever
  int64_t deviceListsMinDraws = 8 << 20;
  int64_t stepsRun = 0;

  double localLoss(const Vec& w, bool test = false) { return test ? model_.loss(w, nTrain_, nRows_) : model_.loss(w, 0, nTrain_); }  // :104
  double localAccuracy(const Vec& w, bool test = false) { return test ? model_.accuracy(w, nTrain_, nRows_) : model_.accuracy(w, 0, nTrain_); }  // :100

  GradState fit(const Vec& initialWeights, int maxEpochs, int batchSize, double learningRate, const EarlyStopping::Criterion& stoppingCriterion) {
    const auto split = SplitStrategy::vanilla(nTrain_, nodeCount_);  // :136
    int64_t maxSamples = 0;                                           // :138
    for (const auto& r : split) maxSamples = std::max(maxSamples, r.second - r.first);
    check(dsgd_set_weights(model_.ctx(), initialWeights.data()));
    GradState state = GradState::start(initialWeights);
    for (int epoch = 0;; ++epoch) {
      const std::optional<double> last = losses.empty() ? std::nullopt : std::optional<double>(losses.front());
      if (epoch >= maxEpochs) return state.finish(last);             // :154 "Reached max number of epochs"
      if (stoppingCriterion(testLosses)) return state.finish(last);   // :166 "Converged to target"
      // :179-199 -- the epoch's batches.  :184: every worker's split is reshuffled for EVERY batch, then sliced; nothing
      // else consumes the generator inside the loop, so the epoch's lists are drawn first (the same draws in the same
      // order) and, with plans, handed over as ONE resident plan: all batches in one launch (5 us per 3 x 100 batch
      // against 40 us per dsgd_sync_step call).
      std::vector<int32_t> flat;
      std::vector<int64_t> offsets{0};
      int64_t nSteps = 0;
      bool emptySlice = false, ranOnDevice = false;
      {
        int64_t nExpected = 0, perBatch = 0;
        for (int64_t batch = 0; batch < maxSamples; batch += batchSize) ++nExpected;
        for (const auto& r : split) perBatch += r.second - r.first - 1;
        if (usePlans && deviceListsMinDraws >= 0 && nExpected * perBatch >= deviceListsMinDraws) {
          std::vector<int64_t> sb, se;
          for (const auto& r : split) {
            sb.push_back(r.first);
            se.push_back(r.second);
          }
          uint64_t st = rnd_.state();
          dsgd_plan* plan = nullptr;
          int64_t n = 0;
          const int rc = dsgd_plan_create_from_seed(model_.ctx(), &st, sb.data(), se.data(), (int32_t)split.size(), maxSamples, batchSize, &plan, &n, nullptr);
          if (rc == DSGD_OK) {
            rnd_.setState(st);
            if (plan) {
              int rr = dsgd_plan_run(model_.ctx(), plan, 0, n, (float)learningRate);
              if (rr == DSGD_OK) rr = dsgd_synchronize(model_.ctx(), nullptr);
              dsgd_plan_destroy(model_.ctx(), plan);
              check(rr);
            }
            nSteps = n;
            emptySlice = n < nExpected;
            ranOnDevice = true;
          } else if (rc != DSGD_EUNSUPPORTED) {
            check(rc);
          }
        }
      }
      for (int64_t batch = 0; !ranOnDevice && batch < maxSamples && !emptySlice; batch += batchSize) {
        for (const auto& r : split)
          if (batch >= r.second - r.first) emptySlice = true;   // the slave would be handed an empty slice: Vec.sum throws there
        if (emptySlice) break;
        for (const auto& r : split) {
          std::vector<int32_t> idx((size_t)(r.second - r.first));
          for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int32_t)(r.first + (int64_t)i);
          idx = shuffle(std::move(idx), rnd_);
          const size_t b = (size_t)std::min<int64_t>(batch, (int64_t)idx.size());
          const size_t e = (size_t)std::min<int64_t>(batch + batchSize, (int64_t)idx.size());
          flat.insert(flat.end(), idx.begin() + (long)b, idx.begin() + (long)e);
          offsets.push_back((int64_t)flat.size());
        }
        ++nSteps;
      }
      const int32_t K = (int32_t)split.size();
      if (ranOnDevice) {
      } else if (usePlans && nSteps > 0) {
        dsgd_plan* plan = nullptr;
        check(dsgd_plan_create_n(model_.ctx(), flat.data(), (int64_t)flat.size(), offsets.data(), nSteps, K, &plan));
        int rc = dsgd_plan_run(model_.ctx(), plan, 0, nSteps, (float)learningRate);
        if (rc == DSGD_OK) rc = dsgd_synchronize(model_.ctx(), nullptr);
        dsgd_plan_destroy(model_.ctx(), plan);
        check(rc);
      } else {
        for (int64_t st = 0; st < nSteps; ++st) {
          std::vector<const int32_t*> ptrs;
          std::vector<int64_t> ns;
          for (int32_t k = 0; k < K; ++k) {
            ptrs.push_back(flat.data() + offsets[(size_t)(st * K + k)]);
            ns.push_back(offsets[(size_t)(st * K + k) + 1] - offsets[(size_t)(st * K + k)]);
          }
          // :186-197 -- gradients of all workers, Vec.mean, w - lr * mean
          check(dsgd_sync_step(model_.ctx(), ptrs.data(), ns.data(), K, (float)learningRate, nullptr));
        }
      }
      stepsRun += nSteps;
      if (emptySlice) throw IllegalArgumentException("requirement failed: Cannot sum an empty list of vectors");  // math/Vec.scala:129
      Vec w(initialWeights.size());
      check(dsgd_get_weights(model_.ctx(), w.data()));
      state = state.replaceGrad(w);
      losses.push_front(localLoss(w));             // :206-209 -- four full passes per epoch
      accs.push_front(localAccuracy(w));
      testLosses.push_front(localLoss(w, true));
      testAccs.push_front(localAccuracy(w, true));
    }
  }

 private:
  SparseSVM& model_;
  int64_t nTrain_, nRows_;
  int nodeCount_;
  JavaRandom rnd_;
};

}  // namespace dsgd
#endif  // DSGD_HPP
