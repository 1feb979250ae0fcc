// C++ host-mirror test (nimfm_amd/host/nimfm.hpp): the reference's own test ideas through the
// compiled-language surface -- tests/test_sgd.nim:16-34 (fitLinear=false => w == 0), :58-89 (warm start),
// :129-151 (score improves) -- on planted synthetic data.  Exit code 0 = all checks passed.
#include <cstdio>
#include <random>

#include "../../nimfm_amd/host/nimfm.hpp"

using namespace nimfm;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

int main() {
  const int64_t n = 200, d = 16;
  const int k = 4;
  std::mt19937_64 g(42);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> data;
  std::vector<int64_t> indices, indptr{0};
  std::vector<std::vector<double>> dense(n, std::vector<double>(d, 0.0));
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t j = 0; j < d; ++j) {
      const double v = U(g);
      if (std::fabs(v) < 0.3) continue;
      dense[i][j] = v; data.push_back(v); indices.push_back(j);
    }
    indptr.push_back((int64_t)data.size());
  }
  // planted degree-2 FM labels
  std::normal_distribution<double> N01(0.0, 1.0);
  std::vector<double> Pt((size_t)k * d), y(n);
  for (auto& v : Pt) v = N01(g);
  for (int64_t i = 0; i < n; ++i) {
    double acc = 0.0;
    for (int s = 0; s < k; ++s) {
      double a1 = 0, a2 = 0;
      for (int64_t j = 0; j < d; ++j) { const double t = Pt[s * d + j] * dense[i][j]; a1 += t; a2 += t * t; }
      acc += 0.5 * (a1 * a1 - a2);
    }
    y[i] = acc;
  }
  CSRDataset X(data, indices, indptr, n, d);

  {  // not fitted -> NotFittedError (model/fm_base.nim:13-15)
    FactorizationMachine fm(regression, 2, k);
    bool threw = false;
    try { fm.decisionFunction(X); } catch (const NotFittedError&) { threw = true; }
    CHECK(threw);
    bool bad = false;
    try { FactorizationMachine f2(regression, 0, k); } catch (const std::invalid_argument&) { bad = true; }
    CHECK(bad);
  }
  {  // score improves (test_sgd.nim:129-151), sequential and mini-batch (maxThreads overload)
    for (int maxThreads : {0, 4}) {
      FactorizationMachine fm(regression, 2, k);
      fm.init(X);
      const double before = fm.score(X, y);
      SGD<Squared> sgd(20, 0.01, 1e-9, 1e-9, 1e-9, Squared(), optimal, 1.0, 0, 0.0);
      sgd.batch = 16;
      sgd.fit(X, y, fm, maxThreads);
      CHECK(fm.score(X, y) < before);
      CHECK(sgd.it == 20 * n + 1);
    }
  }
  {  // the mini-batch rule's knobs (include/nimfm_hip.h: nfm_opt_set_touch_cap, nfm_opt_set_ada_cross, track_viol): a touch cap sums the
     // steps of the samples of a batch that share a coordinate instead of averaging them; the cross products only
     // change AdaGrad's state where a batch touches a coordinate several times; without the stopping criterion's sum the fit is the same
    FactorizationMachine f1(regression, 2, k), f16(regression, 2, k);
    f1.init(X); f16.init(X);
    const double start = f1.score(X, y);
    SGD<Squared> s1(1, 0.01, 1e-9, 1e-9, 1e-9, Squared(), optimal, 1.0, 0, 0.0, false), s16(1, 0.01, 1e-9, 1e-9, 1e-9, Squared(), optimal, 1.0, 0, 0.0, false);
    s1.batch = s16.batch = 64;
    s16.touchCap = 16.0;
    s1.fit(X, y, f1, 4);
    s16.fit(X, y, f16, 4);
    CHECK(f1.score(X, y) < start && f16.score(X, y) < start && f16.P != f1.P);
    FactorizationMachine g0(regression, 2, k), g1(regression, 2, k), g2(regression, 2, k);
    AdaGrad<Squared> a0(2, 0.1, 1e-6, 1e-3, 1e-3, Squared(), 1e-10, 0, 0.0, false), a1(2, 0.1, 1e-6, 1e-3, 1e-3, Squared(), 1e-10, 0, 0.0, false),
        a2(2, 0.1, 1e-6, 1e-3, 1e-3, Squared(), 1e-10, 0, 0.0, false);
    a0.batch = a1.batch = a2.batch = 64;
    a1.adaCross = 0.1;
    a2.trackViol = false;
    a0.fit(X, y, g0, 4);
    a1.fit(X, y, g1, 4);
    a2.fit(X, y, g2, 4);
    CHECK(g0.P != g1.P && g0.P == g2.P && g0.w == g2.w && g0.intercept == g2.intercept);
  }
  {  // fitLinear = false => w stays 0; fitIntercept = false => intercept 0 (test_sgd.nim:16-55)
    FactorizationMachine fm(regression, 2, k, explicit_, false, false);
    SGD<Squared> sgd(5, 0.01, 1e-6, 1e-3, 1e-3, Squared(), optimal, 1.0, 0, 0.0);
    sgd.fit(X, y, fm);
    for (double v : fm.w) CHECK(v == 0.0);
    CHECK(fm.intercept == 0.0);
  }
  {  // warm start: 5 x fit(maxIter=1) == fit(maxIter=5), shuffle off (test_sgd.nim:58-89), SGD and AdaGrad
    FactorizationMachine a(regression, 3, k, explicit_, true, true, true), b(regression, 3, k);
    SGD<Squared> s1(1, 0.01, 1e-6, 1e-3, 1e-3, Squared(), optimal, 1.0, 0, 0.0, false);
    SGD<Squared> s5(5, 0.01, 1e-6, 1e-3, 1e-3, Squared(), optimal, 1.0, 0, 0.0, false);
    for (int r = 0; r < 5; ++r) s1.fit(X, y, a);
    s5.fit(X, y, b);
    CHECK(std::fabs(a.intercept - b.intercept) < 1e-8);
    for (size_t e = 0; e < a.P.size(); ++e) CHECK(std::fabs(a.P[e] - b.P[e]) < 1e-8);
    FactorizationMachine c(regression, 2, k, explicit_, true, true, true), e2(regression, 2, k);
    AdaGrad<Squared> a1(1, 0.1, 1e-6, 1e-3, 1e-3, Squared(), 1e-10, 0, 0.0, false);
    AdaGrad<Squared> a5(5, 0.1, 1e-6, 1e-3, 1e-3, Squared(), 1e-10, 0, 0.0, false);
    for (int r = 0; r < 5; ++r) a1.fit(X, y, c);
    a5.fit(X, y, e2);
    for (size_t e = 0; e < c.P.size(); ++e) CHECK(std::fabs(c.P[e] - e2.P[e]) < 1e-8);
  }
  {  // MBPSGD (optimizer/minibatch_psgd.nim): the loss goes down, a strong L1 penalty empties P, degree 3 is refused
    FactorizationMachine fm(regression, 2, k);
    MBPSGD<Squared, SquaredL12> opt(30, 0.05, 1e-6, 1e-3, 1e-4, 1e-4, Squared(), SquaredL12(), 16, -1, optimal, 1.0, 0, -1.0, false);
    fm.init(X);
    const double before = fm.score(X, y);
    opt.fit(X, y, fm);
    CHECK(fm.score(X, y) < before);
    CHECK(opt.it == 1 + 30 * ((n - 1) / 16 + 1));
    FactorizationMachine z(regression, 2, k);
    MBPSGD<Squared, L1> hard(3, 0.05, 1e-6, 1e-3, 1e-4, 1e3, Squared(), L1(), 16, -1, optimal, 1.0, 0, -1.0, false);
    hard.fit(X, y, z);
    for (double v : z.P) CHECK(v == 0.0);
    FactorizationMachine cubic(regression, 3, k);
    MBPSGD<Squared, SquaredL12> refuse(1, 0.05, 1e-6, 1e-3, 1e-4, 1e-4, Squared(), SquaredL12(), 16, -1, optimal, 1.0, 0);
    bool threw = false;
    try { refuse.fit(X, y, cubic); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
  }
  {  // predictAllWithGrad (optimizer/pgd.nim:70-103): yPred is decisionFunction, the intercept's gradient is mean(dL)
    FactorizationMachine fm(regression, 2, k);
    fm.init(X);
    std::vector<double> yp, dl;
    Grads g = predictAllWithGrad(X, y, fm, yp, dl);
    const std::vector<double> ref = fm.decisionFunction(X);
    double mean_dl = 0.0, mean_loss = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      CHECK(std::fabs(yp[i] - ref[i]) < 1e-12);
      CHECK(std::fabs(dl[i] - (yp[i] - y[i])) < 1e-12);  // Squared.dloss = p - y (loss.nim:21)
      mean_dl += dl[i];
      mean_loss += 0.5 * (y[i] - yp[i]) * (y[i] - yp[i]);
    }
    CHECK(std::fabs(g.intercept - mean_dl / n) < 1e-12);
    CHECK(std::fabs(g.loss - mean_loss / n) < 1e-10);
    CHECK(g.P.size() == fm.P.size() && g.w.size() == (size_t)d);
  }
  {  // loader: dump a small svmlight file (1-based, dumpSVMLightFile's layout), load it on the GPU, same scores
    const char* path = "/tmp/nimfm_host_mirror_test.svm";
    FILE* f = std::fopen(path, "w");
    const int64_t nn = 6;
    const double yy[6] = {1.0, -1.0, 0.5, 2.0, -0.25, 0.0};
    const int64_t ip[7] = {0, 2, 3, 3, 5, 6, 8};
    const int64_t ix[8] = {0, 3, 1, 2, 4, 0, 1, 4};
    const double xv[8] = {0.5, -1.25, 2.0, 0.1, 0.30000000000000004, 1e-3, -7.0, 3.5};
    for (int64_t i = 0; i < nn; ++i) {
      std::fprintf(f, "%.17g", yy[i]);
      for (int64_t q = ip[i]; q < ip[i + 1]; ++q) std::fprintf(f, " %lld:%.17g", (long long)ix[q] + 1, xv[q]);
      if (i + 1 != nn) std::fprintf(f, "\n");
    }
    std::fclose(f);
    std::vector<double> yl;
    auto Xl = loadSVMLightFile(path, yl);
    CHECK(Xl->nSamples() == nn && Xl->nFeatures() == 5);
    for (int64_t i = 0; i < nn; ++i) CHECK(yl[i] == yy[i]);
    CSRDataset Xh(std::vector<double>(xv, xv + 8), std::vector<int64_t>(ix, ix + 8), std::vector<int64_t>(ip, ip + 7), nn, 5);
    FactorizationMachine fm(regression, 2, 3, explicit_, true, true, true);
    fm.init(Xh);
    auto a = fm.decisionFunction(Xh), b = fm.decisionFunction(*Xl);
    for (int64_t i = 0; i < nn; ++i) CHECK(a[i] == b[i]);
    double acc = 0.0;  // score on the device == rmse by hand (metrics.nim:5-13)
    for (int64_t i = 0; i < nn; ++i) acc += (a[i] - yy[i]) * (a[i] - yy[i]);
    CHECK(std::fabs(fm.score(*Xl, yl) - std::sqrt(acc / nn)) < 1e-14);
    std::remove(path);
  }
  std::printf(fails ? "FAILED (%d)\n" : "host mirror ok\n", fails);
  return fails ? 1 : 0;
}
