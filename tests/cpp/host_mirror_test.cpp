// Drives include/dsgd.hpp -- the C++ host mirror of the reference's SparseSVM / Slave / Master surface -- the way the
// reference's own tests would: known answers of java.util.Random, the split / early-stopping rules, and (mode "gpu")
// the two hand-derived known-answer runs of SURVEY.md 8(c) through SparseSVM::gradient, Slave and Master::fit.
//   usage: host_mirror_test cpu | gpu        exit code 0 = all checks passed; "gpu" prints the fit result as JSON
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "dsgd.hpp"

static int failures = 0;
#define CHECK(...)                                                         \
  do {                                                                     \
    if (!(__VA_ARGS__)) {                                                  \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #__VA_ARGS__); \
      ++failures;                                                          \
    }                                                                      \
  } while (0)
static bool close_to(double a, double b, double tol) { return std::fabs(a - b) <= tol; }

static dsgd::Data kat_rows(int n) {  // D = 6, 1-based ids
  dsgd::Data d;
  const std::vector<std::pair<std::vector<std::pair<int32_t, float>>, int>> rows = {
      {{{1, .6f}, {3, .8f}}, +1}, {{{2, 1.f}}, -1}, {{{3, .6f}, {4, .8f}}, -1},
      {{{1, .8f}, {6, .6f}}, +1}, {{{1, .6f}, {3, .8f}}, -1}, {{{2, .6f}, {6, .8f}}, +1}};
  for (int i = 0; i < n; ++i) d.add(rows[(size_t)i].first, rows[(size_t)i].second);
  return d;
}

static void cpu_checks() {
  using namespace dsgd;
  {  // new java.util.Random(0)
    JavaRandom r(0);
    CHECK(r.nextInt() == -1155484576 && r.nextInt() == -723955400 && r.nextInt() == 1033096058);
    JavaRandom q(0);
    const int expect[10] = {0, 8, 9, 7, 5, 3, 1, 1, 9, 4};
    for (int e : expect) CHECK(q.nextInt(10) == e);
    CHECK(JavaRandom(42).nextInt() == -1170105035);
  }
  {  // scala.util.Random.shuffle: a seeded permutation whose LAST element is decided first
    std::vector<int> xs(20);
    for (int i = 0; i < 20; ++i) xs[(size_t)i] = i;
    JavaRandom a(0), b(0), c(0);
    const auto s1 = shuffle(xs, a), s2 = shuffle(xs, b);
    CHECK(s1 == s2 && s1 != xs);
    auto sorted = s1;
    std::sort(sorted.begin(), sorted.end());
    CHECK(sorted == xs && s1[19] == c.nextInt(20));
  }
  {  // SplitStrategy.vanilla: grouped(ceil(n / k)); may yield fewer than k groups (N = 9, K = 4 => 3)
    const auto s = SplitStrategy::vanilla(18519, 3);
    CHECK(s.size() == 3 && s[0].first == 0 && s[0].second == 6173 && s[2].second == 18519);
    CHECK(SplitStrategy::vanilla(9, 4).size() == 3);
  }
  {  // EarlyStopping (newest first)
    const auto t = EarlyStopping::target(0.5);
    CHECK(!t({}) && t({0.4, 0.9}) && !t({0.6, 0.4}));
    const auto ni = EarlyStopping::noImprovement(2, 0.01);
    CHECK(!ni({}));
    CHECK(!ni({0.5, 0.6, 0.7}));        // the newest is the best
    CHECK(ni({0.7, 0.6, 0.5, 0.9}));    // the best is 2 = patience steps old
    CHECK(!ni({0.7, 0.5, 0.9}));        // ... only 1 step old
    CHECK(!EarlyStopping::noImprovement(2, 0.01, 2)({0.7, 0.6, 0.5, 0.9}));  // minSteps < size => false (EarlyStopping.scala:44)
  }
  {  // GradState
    auto g = GradState::start(Vec{0, 1}).replaceGrad(Vec{0, 2}).finish(0.25);
    CHECK(g.updates == 1 && g.finished && g.loss && *g.loss == 0.25 && g.grad[1] == 2);
  }
  {  // errors map to the exceptions the reference throws; no device => no silent CPU fallback
    bool threw = false;
    try {
      dsgd_config cfg{};
      dsgd_ctx* c = nullptr;
      check(dsgd_create(&cfg, &c));  // n_features = 0
    } catch (const IllegalArgumentException&) {
      threw = true;
    }
    CHECK(threw);
    if (dsgd_device_count() == 0) {
      threw = false;
      try {
        SparseSVM model(0.1, 6);
      } catch (const NativeError& e) {
        threw = e.code == DSGD_EUNSUPPORTED;
      }
      CHECK(threw);
    }
  }
}

static void gpu_checks() {
  using namespace dsgd;
  {  // KAT-1: lambda 0.1, lr 0.25, two workers with fixed batches
    SparseSVM model(0.1, 6);
    model.load(kat_rows(6));
    const Vec ds = model.buildDimSparsity(6);
    CHECK(close_to(ds[0], .25, 1e-7) && close_to(ds[1], 1. / 3, 1e-7) && close_to(ds[2], .25, 1e-7) && close_to(ds[3], .5, 1e-7) &&
          ds[4] == 0 && close_to(ds[5], 1. / 3, 1e-7));
    Slave slave(model, /*async=*/false);
    Vec w(7, 0.f);
    CHECK(close_to(model.loss(w, 0, 6), 1.0, 1e-7) && model.accuracy(w, 0, 6) == 0.0);
    const double g0[3][7] = {{0, .6, -1, .2, -.8, 0, 0}, {0, .6 + 1. / 300, 0, .8 + 1. / 300, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0}};
    const double loss_after[3] = {0.339208333, 0.339212638, 0.340734158};
    const double acc_after[3] = {4. / 6, 5. / 6, 5. / 6};
    for (int step = 0; step < 3; ++step) {
      const GradUpdate a = slave.gradient(GradientRequest{w, {0, 1, 2}});
      const GradUpdate b = slave.gradient(GradientRequest{w, {3, 4, 5}});
      for (int j = 0; j < 7; ++j) CHECK(close_to(a.gradUpdate[(size_t)j], g0[step][j], 2e-6));
      for (int j = 0; j < 7; ++j) w[(size_t)j] -= 0.25f * 0.5f * (a.gradUpdate[(size_t)j] + b.gradUpdate[(size_t)j]);  // Vec.mean, w - lr * grad
      CHECK(close_to(model.loss(w, 0, 6), loss_after[step], 1e-6));
      CHECK(close_to(model.accuracy(w, 0, 6), acc_after[step], 1e-9));
    }
    const ForwardReply p = slave.forward(ForwardRequest{{0, 1, 2, 3, 4, 5}, w});
    int correct = 0;
    const int y[6] = {1, -1, -1, 1, -1, 1};
    for (int i = 0; i < 6; ++i) correct += p.predictions[(size_t)i] == (float)y[i];
    CHECK(correct == 5);
    // what makes the reference throw
    bool threw = false;
    try {
      slave.gradient(GradientRequest{w, {}});  // Vec.sum of an empty batch
    } catch (const IllegalArgumentException&) {
      threw = true;
    }
    CHECK(threw);
    threw = false;
    try {
      slave.gradient(GradientRequest{w, {6}});  // data(6) of a 6-row array
    } catch (const IndexOutOfBoundsException&) {
      threw = true;
    }
    CHECK(threw);
    threw = false;
    try {
      slave.stopAsync();  // synchronous-mode slave
    } catch (const IllegalArgumentException&) {
      threw = true;
    }
    CHECK(threw);
  }
  {  // KAT-2: after one step every row is inactive: empty gradients, weights unchanged
    SparseSVM model(0.1, 6);
    model.load(kat_rows(4));
    model.buildDimSparsity(4);
    Slave slave(model, false);
    Vec w(7, 0.f);
    const GradUpdate a = slave.gradient(GradientRequest{w, {0, 1}}), b = slave.gradient(GradientRequest{w, {2, 3}});
    for (int j = 0; j < 7; ++j) w[(size_t)j] -= 0.5f * 0.5f * (a.gradUpdate[(size_t)j] + b.gradUpdate[(size_t)j]);
    const double w1[7] = {0, -.35, .25, -.05, .2, 0, -.15};
    for (int j = 0; j < 7; ++j) CHECK(close_to(w[(size_t)j], w1[j], 1e-6));
    dsgd_batch_stats st{};
    const Vec g = model.gradient(w, {0, 1}, &st);
    bool any = false;
    for (float v : g) any = any || v != 0.f;
    CHECK(!any && st.n_active == 0 && st.n_samples == 2);
  }
  {  // Master.fit over the hosted workers: 2 epochs, 2 workers, batch 2, the reference's RNG stream
    SparseSVM model(0.1, 6);
    model.load(kat_rows(6));
    model.buildDimSparsity(4);
    Master master(model, /*nTrain=*/4, /*nRows=*/6, /*nodeCount=*/2, JavaRandom(0));
    const GradState s = master.fit(Vec(7, 0.f), /*maxEpochs=*/2, /*batchSize=*/2, /*learningRate=*/0.25, EarlyStopping::noImprovement(5, 0.01));
    CHECK(s.finished && s.updates == 2 && master.losses.size() == 2 && master.testLosses.size() == 2 && s.loss && *s.loss == master.losses.front());
    std::printf("{\"updates\": %lld, \"weights\": [", (long long)s.updates);
    for (size_t j = 0; j < s.grad.size(); ++j) std::printf("%s%.9g", j ? ", " : "", (double)s.grad[j]);
    std::printf("], \"losses\": [%.9g, %.9g], \"test_losses\": [%.9g, %.9g]}\n", master.losses[0], master.losses[1], master.testLosses[0],
                master.testLosses[1]);
    // the same fit with one request per batch (usePlans = false): the same stream, the same lists, the same weights
    SparseSVM model2(0.1, 6);
    model2.load(kat_rows(6));
    model2.buildDimSparsity(4);
    Master master2(model2, 4, 6, 2, JavaRandom(0));
    master2.usePlans = false;
    const GradState s2 = master2.fit(Vec(7, 0.f), 2, 2, 0.25, EarlyStopping::noImprovement(5, 0.01));
    CHECK(master.stepsRun == master2.stepsRun && master.stepsRun == 2);
    for (size_t j = 0; j < s.grad.size(); ++j) CHECK(close_to(s.grad[j], s2.grad[j], 1e-7));
    // ... and with the epoch's lists DRAWN BY THE DEVICE (dsgd_plan_create_from_seed, forced for this tiny epoch): the same
    // stream, so the same lists, the same weights bit for bit as the plan made from the host's draws
    SparseSVM model3(0.1, 6);
    model3.load(kat_rows(6));
    model3.buildDimSparsity(4);
    Master master3(model3, 4, 6, 2, JavaRandom(0));
    master3.deviceListsMinDraws = 0;
    const GradState s3 = master3.fit(Vec(7, 0.f), 2, 2, 0.25, EarlyStopping::noImprovement(5, 0.01));
    CHECK(master3.stepsRun == 2 && s3.updates == s.updates);
    for (size_t j = 0; j < s.grad.size(); ++j) CHECK(s.grad[j] == s3.grad[j]);
  }
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  cpu_checks();
  if (mode == "gpu") gpu_checks();
  if (failures) {
    std::fprintf(stderr, "%d check(s) failed\n", failures);
    return 1;
  }
  std::fprintf(stderr, "host mirror (%s): all checks passed\n", mode.c_str());
  return 0;
}
