#include "lc_topic.hpp"

#include "lc_engine.hpp"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <limits>
#include <stdexcept>

namespace lce {

using lch::GaussWishState;
using lch::WeightState;

namespace {

struct GreedOrder {  // src/comutils.h:44-49
  int k;
  int tally;
  double Fk;
};
bool greedcomp(const GreedOrder& i, const GreedOrder& j) {  // src/comutils.h:60-68
  if (i.tally == j.tally) return i.Fk > j.Fk;
  return i.tally < j.tally;
}

bool anyempty(const std::vector<GaussWishState>& c) {  // src/comutils.h:114-123
  for (const auto& x : c)
    if (x.N <= 1) return true;
  return false;
}

// probutils.cpp:141-150, one row
double logsumexp_row(const double* v, int n) {
  double mx = v[0];
  for (int i = 1; i < n; ++i) mx = std::max(mx, v[i]);
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += std::exp(v[i] - mx);
  return std::log(s) + mx;
}

void resize_weights(std::vector<WeightState>& w, int n, int kind, double prior) {
  if ((int)w.size() > n) w.resize(n);
  while ((int)w.size() < n) w.emplace_back(kind, prior);
}

}  // namespace

// ---------------------------------------------------------------------------
// scluster.cpp:172-260 / mcluster.cpp:186-287
// ---------------------------------------------------------------------------
double topic_vbem(lcc::Context& ctx, const TopicData& data, TopicModel& model, const TopicOptions& opt) {
  const int I = ctx.J(), K = ctx.K(), D = ctx.D(), T = model.T, J = data.J, Dt = data.Dt;
  const bool mcm = data.W != nullptr;
  if (I != data.Itot) throw std::invalid_argument("the context must hold one group per document");
  if (K < 1 || T < 1) throw std::invalid_argument("qZ and qY must have at least one column");
  if ((int64_t)model.qY.size() != (int64_t)I * T) throw std::invalid_argument("qY has the wrong size");

  // weights_j.resize(J, WJ()); weights_t.resize(T, WT(prior_t) | WT()); clusters(_t).resize(...)   (:189-192 / :205-209)
  resize_weights(model.weights_j, J, lch::W_GDIRICHLET, lch::ALPHA1PRIOR);
  resize_weights(model.weights_t, T, lch::W_DIRICHLET, mcm ? lch::ALPHA1PRIOR : opt.prior_t);
  if (mcm) {
    if ((int)model.clusters_t.size() > T) model.clusters_t.resize(T);
    while ((int)model.clusters_t.size() < T) model.clusters_t.emplace_back(opt.prior_t, Dt);
  }
  if ((int)model.clusters.size() > K) model.clusters.resize(K);
  while ((int)model.clusters.size() < K) model.clusters.emplace_back(opt.prior_k, D);

  std::vector<double> Nk(K), xs((size_t)K * D), xxs((size_t)K * D * D), Njik((size_t)I * K);
  std::vector<double> Ntk((size_t)T * K), qysum((size_t)J * T);
  std::vector<double>&A = model.lastA, &m = model.lastm, &c = model.lastc, &cst = model.cst;
  A.assign((size_t)K * D * D, 0.0);
  m.assign((size_t)K * D, 0.0);
  c.assign((size_t)I * K, 0.0);
  cst.assign(K, 0.0);
  std::vector<double> part, fpart;  // per-chunk partial sums of the document loops
  std::vector<double> At, ctt, wN(T), wxs, wxx, ellw;  // MCM top-level Gaussians
  if (mcm) {
    At.resize((size_t)T * Dt * Dt);
    ctt.resize(T);
    wxs.resize((size_t)T * Dt);
    wxx.resize((size_t)T * Dt * Dt);
    ellw.resize((size_t)I * T);
  }
  std::vector<double>& qY = model.qY;

  double F = std::numeric_limits<double>::max(), Fold;
  int it = 0, done = 0;
  bool again;
  do {
    Fold = F;
    // bottom-level statistics of ALL documents in one pass; N_jik = per-document column sums (:62-63, :199-203)
    ctx.suffstat(nullptr, Nk.data(), xs.data(), xxs.data(), Njik.data());

    // Ntk = sum_ji qY_ji^T N_jik; weights_j[j].update(qY[j].colwise().sum())   (:199-207 / :221-233)
    // The document loops run in NCH fixed chunks on the worker pool; chunk results are folded in chunk order, so the
    // sums do not depend on the number of threads.
    std::fill(Ntk.begin(), Ntk.end(), 0.0);
    std::fill(qysum.begin(), qysum.end(), 0.0);
    const int NCH = std::min(I, 64), per = (I + NCH - 1) / NCH;
    const double docwork = 4.0 * T * K;
    part.assign((size_t)NCH * T * K, 0.0);
    parallel_chunks(NCH, opt.nthreads, docwork * per, [&](int ch) {
      double* nt = part.data() + (size_t)ch * T * K;
      for (int i = ch * per; i < std::min(I, (ch + 1) * per); ++i) {
        const double* qy = qY.data() + (size_t)i * T;
        const double* nk = Njik.data() + (size_t)i * K;
        for (int t = 0; t < T; ++t)
          for (int k = 0; k < K; ++k) nt[(size_t)t * K + k] += qy[t] * nk[k];
      }
    });
    for (int ch = 0; ch < NCH; ++ch)
      for (int e = 0; e < T * K; ++e) Ntk[(size_t)e] += part[(size_t)ch * T * K + e];
    for (int i = 0; i < I; ++i)
      for (int t = 0; t < T; ++t) qysum[(size_t)data.doc_group[i] * T + t] += qY[(size_t)i * T + t];
    if (ctx.group_sharded()) ctx.allreduce_values(Ntk.data(), T * K);  // groups (and their documents) are sharded
    for (int j = 0; j < J; ++j) model.weights_j[j].update(qysum.data() + (size_t)j * T, T);

    // VBM for the top-level cluster parameters (:210-212 / :236-246)
    if (mcm) {
      std::fill(wN.begin(), wN.end(), 0.0);
      std::fill(wxs.begin(), wxs.end(), 0.0);
      std::fill(wxx.begin(), wxx.end(), 0.0);
      for (int i = 0; i < I; ++i) {
        const double* w = data.W + (size_t)i * Dt;
        for (int t = 0; t < T; ++t) {
          const double q = qY[(size_t)i * T + t];
          wN[t] += q;
          for (int a = 0; a < Dt; ++a) {
            const double qa = q * w[a];
            wxs[(size_t)t * Dt + a] += qa;
            for (int b = 0; b < Dt; ++b) wxx[((size_t)t * Dt + a) * Dt + b] += qa * w[b];
          }
        }
      }
      if (ctx.group_sharded()) {
        ctx.allreduce_values(wN.data(), T);
        ctx.allreduce_values(wxs.data(), T * Dt);
        ctx.allreduce_values(wxx.data(), T * Dt * Dt);
      }
    }
    for (int t = 0; t < T; ++t) {
      if (mcm) {
        GaussWishState& ct = model.clusters_t[t];
        ct.clearobs();
        ct.addstats(wN[t], wxs.data() + (size_t)t * Dt, wxx.data() + (size_t)t * Dt * Dt);
      }
      model.weights_t[t].update(Ntk.data() + (size_t)t * K, K);
      if (mcm) {
        GaussWishState& ct = model.clusters_t[t];
        ct.update();
        const std::vector<double> a = ct.whitener();
        std::copy(a.begin(), a.end(), At.begin() + (size_t)t * Dt * Dt);
        ctt[t] = ct.eloglike_const();
      }
    }

    // VBM for the bottom-level clusters (:215-225 / :249-258)
    for (int k = 0; k < K; ++k) {
      GaussWishState& cl = model.clusters[k];
      cl.clearobs();
      cl.addstats(Nk[k], xs.data() + (size_t)k * D, xxs.data() + (size_t)k * D * D);
      cl.update();
      const std::vector<double> Ak = cl.whitener();
      std::copy(Ak.begin(), Ak.end(), A.begin() + (size_t)k * D * D);
      std::copy(cl.m.begin(), cl.m.end(), m.begin() + (size_t)k * D);
      cst[k] = cl.eloglike_const();
    }

    // VBE for the top-level indicators, vbeY (scluster.cpp:50-85 / mcluster.cpp:49-92), with the OLD N_jik
    double Fyz = 0.0;
    if (mcm) {  // GaussWish::Eloglike of the document observations (distributions.cpp:356-370)
      std::vector<double> y(Dt);
      for (int i = 0; i < I; ++i) {
        const double* w = data.W + (size_t)i * Dt;
        for (int t = 0; t < T; ++t) {
          const double* a = At.data() + (size_t)t * Dt * Dt;
          const std::vector<double>& mt = model.clusters_t[t].m;
          double d2 = 0.0;
          for (int r = 0; r < Dt; ++r) {
            double s = 0.0;
            for (int cc = 0; cc <= r; ++cc) s += a[(size_t)r * Dt + cc] * (w[cc] - mt[cc]);
            d2 += s * s;
          }
          ellw[(size_t)i * T + t] = ctt[t] - 0.5 * d2;
        }
      }
    }
    fpart.assign((size_t)NCH, 0.0);
    parallel_chunks(NCH, opt.nthreads, docwork * per, [&](int ch) {
      std::vector<double> like((size_t)T), logq((size_t)T);
      double f = 0.0;
      for (int i = ch * per; i < std::min(I, (ch + 1) * per); ++i) {
        const WeightState& wj = model.weights_j[data.doc_group[i]];
        const double* nk = Njik.data() + (size_t)i * K;
        for (int t = 0; t < T; ++t) {
          const std::vector<double>& el = model.weights_t[t].Elogpi;
          double s = 0.0;
          for (int k = 0; k < K; ++k) s += nk[k] * el[k];
          like[t] = s;
          logq[t] = mcm ? s + wj.Elogpi[t] + ellw[(size_t)i * T + t] : wj.Elogpi[t] + s;
        }
        const double logZ = logsumexp_row(logq.data(), T);
        double acc = 0.0;
        for (int t = 0; t < T; ++t) {
          const double q = std::exp(logq[t] - logZ);
          qY[(size_t)i * T + t] = q;
          acc += q * like[t];
        }
        f += acc - logZ;
      }
      fpart[(size_t)ch] = f;
    });
    for (int ch = 0; ch < NCH; ++ch) Fyz += fpart[(size_t)ch];
    if (ctx.group_sharded()) Fyz = ctx.allreduce_value(Fyz);

    // VBE for the bottom-level indicators, vbeZ, with the NEW qY: one E-step launch over all documents
    parallel_chunks(NCH, opt.nthreads, docwork * per, [&](int ch) {
      std::vector<double> Et((size_t)K);
      for (int i = ch * per; i < std::min(I, (ch + 1) * per); ++i) {
        std::fill(Et.begin(), Et.end(), 0.0);
        for (int t = 0; t < T; ++t) {
          const double q = qY[(size_t)i * T + t];
          const std::vector<double>& el = model.weights_t[t].Elogpi;
          for (int k = 0; k < K; ++k) Et[(size_t)k] += q * el[k];
        }
        for (int k = 0; k < K; ++k) c[(size_t)i * K + k] = Et[(size_t)k] + cst[k];
      }
    });
    double Fz = 0.0;
    ctx.estep(K, A.data(), m.data(), c.data(), &Fz, nullptr);

    // fenergy (scluster.cpp:131-160 / mcluster.cpp:142-175)
    double Fw = 0.0, Fc = 0.0, Fk = 0.0;
    for (int j = 0; j < J; ++j) Fw += model.weights_j[j].fenergy();
    if (ctx.group_sharded()) Fw = ctx.allreduce_value(Fw);
    for (int t = 0; t < T; ++t) Fc += model.weights_t[t].fenergy() + (mcm ? model.clusters_t[t].fenergy() : 0.0);
    for (int k = 0; k < K; ++k) Fk += model.clusters[k].fenergy();
    F = Fw + Fc + Fk + Fyz + Fz;
    if (opt.trace) opt.trace->push_back(F);

    if (opt.fixed_iters >= 0) {
      again = ++done < opt.fixed_iters;
      continue;
    }
    if ((F - Fold) / std::abs(Fold) > lch::FENGYDEL) throw std::runtime_error("Free energy increase!");
    if (opt.verbose) std::cout << '-' << std::flush;
    again = (std::abs((Fold - F) / Fold) > lch::CONVERGE) && ((++it < opt.maxit) || (opt.maxit < 0));
  } while (again);
  return F;
}

// ---------------------------------------------------------------------------
// scluster.cpp:280-428 (split_gr) / mcluster.cpp:309-455 (ssplit)
// ---------------------------------------------------------------------------
static bool topic_split(lcc::Context& ctx, const TopicData& data, TopicModel& model, std::vector<int>& tally,
                        double F, const TopicOptions& opt) {
  const int I = ctx.J(), K = (int)model.clusters.size(), D = ctx.D(), T = model.T;
  const bool mcm = data.W != nullptr;
  if (K >= opt.maxK && opt.maxK >= 0) return false;
  tally.resize(K, 0);

  // ord[k].Fk = fenergy_k - sum_ji q_jik . Eloglike_k(X_ji)   (:303-325 / :332-354): the data term is
  // const_k * sum q_k + LL_k with LL_k from a repeat of the last E-step (same parameters => same qZ)
  std::vector<double> LLk(K), Njik((size_t)I * K), Nq(K, 0.0);
  double Fz = 0.0;
  ctx.estep(K, model.lastA.data(), model.lastm.data(), model.lastc.data(), &Fz, LLk.data());
  ctx.colsums(Njik.data());
  for (int i = 0; i < I; ++i)
    for (int k = 0; k < K; ++k) Nq[k] += Njik[(size_t)i * K + k];
  if (ctx.group_sharded()) ctx.allreduce_values(Nq.data(), K);
  std::vector<GreedOrder> ord(K);
  for (int k = 0; k < K; ++k) {
    ord[k].k = k;
    ord[k].tally = tally[k];
    ord[k].Fk = model.clusters[k].fenergy() - (model.cst[k] * Nq[k] + LLk[k]);
  }
  std::sort(ord.begin(), ord.end(), greedcomp);

  const double prior_k = model.clusters[0].prior;
  const double prior_t = mcm ? model.clusters_t[0].prior : opt.prior_t;
  lcc::RowSelection sel;
  std::vector<double> njs, eigv;

  for (const GreedOrder& o : ord) {
    const int k = o.k;
    ++tally[k];
    if (model.clusters[k].N < 4) continue;

    // partobs + splitobs per document (:344-374 / :371-403) on the device
    ctx.select_rows(k, 0.5, sel);
    const int64_t Mtot = (int64_t)std::llround(ctx.allreduce_value((double)sel.M));
    lcc::Context sub(ctx.device(), ctx.stream());
    sub.inherit_comm(ctx);
    sub.set_data_gather(ctx, sel);
    lch::eigpower(model.clusters[k].iW, D, eigv);
    sub.qz_init_split(model.clusters[k].m.data(), eigv.data());
    njs.assign((size_t)I * 2, 0.0);
    sub.colsums(njs.data());
    double sc = 0.0;
    for (int i = 0; i < I; ++i) sc += njs[(size_t)i * 2];
    if (ctx.group_sharded()) sc = ctx.allreduce_value(sc);
    const int64_t scount = (int64_t)std::llround(sc);
    if (scount < 2 || scount > Mtot - 2) continue;

    // refine the split (:377-386 / :410-416): SCM with ONE top-level cluster, MCM with the current qY and W
    TopicModel ms;
    TopicOptions so;
    so.prior_t = prior_t;
    so.prior_k = prior_k;
    so.maxit = (int)lch::SPLITITER;
    so.nthreads = opt.nthreads;
    if (mcm) {
      ms.T = T;
      ms.qY = model.qY;
    } else {
      ms.T = 1;
      ms.qY.assign((size_t)I, 1.0);
    }
    topic_vbem(sub, data, ms, so);
    if (anyempty(ms.clusters)) continue;

    // auglabels (:392-398 / :422-428) on a copy of qZ, then one iteration with ALL data and a copy of qY
    ctx.qz_clone_to_alt();
    ctx.qz_swap_alt();
    ctx.qz_split_from(sub, sel, k);
    ms.T = T;
    ms.qY = model.qY;
    so.maxit = 1;
    double Fs;
    try {
      Fs = topic_vbem(ctx, data, ms, so);
    } catch (...) {
      ctx.qz_swap_alt();
      throw;
    }
    if (anyempty(ms.clusters)) {
      ctx.qz_swap_alt();
      continue;
    }
    if (opt.verbose) std::cout << '=' << std::flush;
    if ((Fs < F) && (std::abs((F - Fs) / F) > lch::CONVERGE)) {
      model.qY = ms.qY;  // qY = qYaug; qZ = qZaug (the augmented buffer is now the current one)
      tally[k] = 0;
      return true;
    }
    ctx.qz_swap_alt();
  }
  return false;
}

// scluster.cpp:437-481 / mcluster.cpp:465-511
static bool prune_clusters_t(TopicModel& model, int I, bool mcm, bool verbose) {
  const int T = model.T;
  std::vector<int> keep;
  for (int t = 0; t < T; ++t) {
    double n = 0.0;
    for (double v : model.weights_t[t].Nk) n += v;
    if (!(n < 1)) keep.push_back(t);
  }
  if ((int)keep.size() == T) return false;
  if (verbose) std::cout << '*' << std::flush;
  std::vector<WeightState> nw;
  std::vector<GaussWishState> nc;
  for (int t : keep) {
    nw.push_back(std::move(model.weights_t[t]));
    if (mcm) nc.push_back(std::move(model.clusters_t[t]));
  }
  model.weights_t.swap(nw);
  if (mcm) model.clusters_t.swap(nc);
  const int nT = (int)keep.size();
  std::vector<double> q((size_t)I * nT);
  for (int i = 0; i < I; ++i)
    for (int t = 0; t < nT; ++t) q[(size_t)i * nT + t] = model.qY[(size_t)i * T + keep[t]];
  model.qY.swap(q);
  model.T = nT;
  return true;
}

// ---------------------------------------------------------------------------
// scluster.cpp:493-570 / mcluster.cpp:525-605
// ---------------------------------------------------------------------------
double topic_cluster(lcc::Context& ctx, const TopicData& data, TopicModel& model, const TopicOptions& opt,
                     std::vector<TopicRound>* rounds) {
  if (opt.nthreads < 1) throw std::invalid_argument("Must specify at least one thread for execution!");
  const bool mcm = data.W != nullptr;
  ctx.qz_fill(1, 1.0);
  bool issplit = true, emptyclasses = true;
  double F = 0.0;
  std::vector<int> tally;
  while (issplit || emptyclasses) {
    std::vector<double> tr;
    TopicOptions vo = opt;
    vo.maxit = -1;
    vo.trace = &tr;
    F = topic_vbem(ctx, data, model, vo);
    if (rounds) rounds->push_back(TopicRound{model.T, (int)model.clusters.size(), tr});
    if (opt.verbose) std::cout << '<' << std::flush;
    if (!issplit)
      emptyclasses = prune_clusters_t(model, data.Itot, mcm, opt.verbose);
    else
      issplit = topic_split(ctx, data, model, tally, F, opt);
    if (opt.verbose) std::cout << '>' << std::endl;
  }
  if (opt.verbose) {
    std::cout << "Finished!" << std::endl;
    std::cout << "Number of top level clusters = " << model.T;
    std::cout << ", and bottom level clusters = " << model.clusters.size() << std::endl;
    std::cout << "Free energy = " << F << std::endl;
  }
  return F;
}

}  // namespace lce
