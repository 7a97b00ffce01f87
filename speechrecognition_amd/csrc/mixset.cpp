// mixset.cpp -- host-side model loading: MIXSET v2 file -> finalised per-density tables -> sr_model.
//
// Mirrors what the reference does once at start-up, MixtureModel::MixtureModel with
// "load-mixtures-from" (sietill/Mixtures.cpp:156-174) -> read() (:748-830) -> finalize() (:374-461,
// calculate_variance :251-275).  The arithmetic below keeps the reference's operation order
// (acc / weight, then  - mean*mean, 1 / var, left-to-right log sum) so that the tables, and with
// them the SR_GMM_EXACT scores, are bit-identical to the reference's private means_/vars_inv_/
// norm_/mean_weights_log_.  One-off work, stays on the host like in the reference.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/srgpu.h"
#include "host_util.h"

namespace {

struct Reader {
  FILE* f;
  explicit Reader(const char* path) : f(fopen(path, "rb")) {}
  ~Reader() { if (f) fclose(f); }
  bool get(void* dst, size_t n) { return fread(dst, 1, n, f) == n; }
  template <typename T> bool get(T* v) { return get(v, sizeof(T)); }
};

struct Accumulators {  // one block of read_accumulator (Mixtures.cpp:104-129)
  std::vector<double> sum;     // [n x dim]
  std::vector<double> weight;  // [n]
  uint32_t n = 0;
};

const char* read_block(Reader& r, uint32_t dim, Accumulators* a) {
  if (!r.get(&a->n)) return "Error reading size";
  a->sum.resize((size_t)a->n * dim);
  a->weight.resize(a->n);
  for (uint32_t i = 0; i < a->n; i++) {
    uint32_t d;
    if (!r.get(&d)) return "Error reading dimension";
    if (d != dim) return "Invalid dimension";
    if (!r.get(a->sum.data() + (size_t)i * dim, sizeof(double) * dim)) return "Error reading features";
    if (!r.get(&a->weight[i])) return "Error reading weight";
  }
  return nullptr;
}

struct Density { uint32_t mean, var; };

}  // namespace

namespace srhost {

const char* load_mixset(const char* path, uint32_t dim, int pooling, MixsetTables* out) {
  Reader r(path);
  if (!r.f) return "cannot open model file";
  char magic[8];
  static const char kMagic[8] = {'M', 'I', 'X', 'S', 'E', 'T', 0, 0};
  uint32_t version = 0, fdim = 0;
  if (!r.get(magic, 8)) return "Error reading magic header";
  if (memcmp(magic, kMagic, 8) != 0) return "Invalid magic header";
  if (!r.get(&version)) return "Error reading version";
  if (version != 2u) return "Invalid version";
  if (!r.get(&fdim)) return "Error reading dimension";
  if (fdim != dim) return "Invalid dimension";
  Accumulators mean_acc, var_acc;
  if (const char* e = read_block(r, dim, &mean_acc)) return e;
  if (const char* e = read_block(r, dim, &var_acc)) return e;
  uint32_t n_dens = 0;
  if (!r.get(&n_dens)) return "Error reading density count";
  std::vector<Density> dens(n_dens);
  for (auto& d : dens) {
    if (!r.get(&d.mean)) return "Error reading mean_idx";
    if (d.mean >= mean_acc.n) return "Invalid mean_idx";
    if (!r.get(&d.var)) return "Error reading var_idx";
    if (d.var >= var_acc.n) return "Invalid var_idx";
  }
  uint32_t n_mix = 0;
  if (!r.get(&n_mix)) return "Error reading mixture count";
  std::vector<std::vector<Density>> mixtures(n_mix);
  for (auto& mix : mixtures) {
    uint32_t nd = 0;
    if (!r.get(&nd)) return "Error reading density count for mixture";
    mix.reserve(nd);
    for (uint32_t i = 0; i < nd; i++) {
      uint32_t di;
      double w;
      if (!r.get(&di)) return "Error reading density idx";
      if (di >= n_dens) return "Invalid density idx";
      if (!r.get(&w)) return "Error reading density weight";
      if (w != mean_acc.weight[dens[di].mean]) return "Inconsistent density weight";  // Mixtures.cpp:825
      mix.push_back(dens[di]);
    }
  }

  // ---- finalize ----------------------------------------------------------------------------------
  const size_t D = dim;
  std::vector<double> mean(mean_acc.sum.size()), logw(mean_acc.n, 0.0);
  std::vector<double> var(var_acc.sum.size(), 0.0), ivar(var_acc.sum.size(), 0.0), nrm(var_acc.n, 0.0);
  auto variance = [&](uint32_t vi, const double* mu) {  // calculate_variance
    double* v = &var[vi * D];
    for (size_t d = 0; d < D; d++) v[d] = var_acc.sum[vi * D + d] / var_acc.weight[vi];
    for (size_t d = 0; d < D; d++) v[d] = v[d] - mu[d] * mu[d];
    for (size_t d = 0; d < D; d++) ivar[vi * D + d] = 1 / v[d];
    double acc = D * log(2 * M_PI);
    for (size_t d = 0; d < D; d++) acc = acc + log(v[d]);
    nrm[vi] = acc / 2;
  };
  double total = 0.0;
  std::vector<double> pooled(D);
  for (auto& mix : mixtures) {
    double mix_total = 0.0;
    for (const Density& dn : mix) {
      mix_total += mean_acc.weight[dn.mean];
      for (size_t d = 0; d < D; d++) mean[dn.mean * D + d] = mean_acc.sum[dn.mean * D + d] / mean_acc.weight[dn.mean];
      if (pooling == SRHOST_POOL_NONE) variance(dn.var, &mean[dn.mean * D]);
    }
    for (const Density& dn : mix) logw[dn.mean] = log(mean_acc.weight[dn.mean] / mix_total);
    if (pooling == SRHOST_POOL_MIXTURE && !mix.empty()) {
      std::fill(pooled.begin(), pooled.end(), 0.0);
      for (const Density& dn : mix)
        for (size_t d = 0; d < D; d++) pooled[d] = pooled[d] + mean_acc.sum[dn.mean * D + d];
      for (size_t d = 0; d < D; d++) pooled[d] = pooled[d] / mix_total;
      variance(mix[0].var, pooled.data());
    }
    total += mix_total;
  }
  if (pooling == SRHOST_POOL_GLOBAL) {
    std::fill(pooled.begin(), pooled.end(), 0.0);
    for (auto& mix : mixtures)
      for (const Density& dn : mix)
        for (size_t d = 0; d < D; d++) pooled[d] = pooled[d] + mean_acc.sum[dn.mean * D + d];
    for (size_t d = 0; d < D; d++) pooled[d] = pooled[d] / total;
    variance(0, pooled.data());
  }

  // ---- expand per density in mixture order ----------------------------------------------------------
  out->dim = dim;
  out->dens_off.assign(1, 0u);
  out->means.clear(); out->inv_vars.clear(); out->norm.clear(); out->logw.clear();
  out->dens_mean.clear(); out->dens_var.clear();
  out->n_mean = mean_acc.n; out->n_var = var_acc.n;
  for (auto& mix : mixtures) {
    for (const Density& dn : mix) {
      out->means.insert(out->means.end(), &mean[dn.mean * D], &mean[dn.mean * D] + D);
      out->inv_vars.insert(out->inv_vars.end(), &ivar[dn.var * D], &ivar[dn.var * D] + D);
      out->norm.push_back(nrm[dn.var]);
      out->logw.push_back(logw[dn.mean]);
      out->dens_mean.push_back(dn.mean);
      out->dens_var.push_back(dn.var);
    }
    out->dens_off.push_back((uint32_t)out->norm.size());
  }
  return nullptr;
}

}  // namespace srhost

extern "C" SR_API int sr_model_load_mixset(const char* path, uint32_t dim, int pooling, int max_approx, int device,
                                           sr_model** out) {
  if (!path || !out) return srhost::set_error(SR_EINVAL, "null argument");
  if (pooling < 0 || pooling > 2) return srhost::set_error(SR_EINVAL, "pooling must be 0 (global), 1 (mixture) or 2 (none)");
  srhost::MixsetTables t;
  if (const char* e = srhost::load_mixset(path, dim, pooling, &t)) return srhost::set_error(SR_EINVAL, e);
  int rc = sr_model_create(device, dim, (uint32_t)t.dens_off.size() - 1, t.dens_off.data(), t.means.data(), t.inv_vars.data(),
                           t.norm.data(), t.logw.data(), max_approx, out);
  if (rc == SR_OK && !t.dens_mean.empty()) {
    rc = sr_model_set_tying(*out, t.n_mean, t.n_var, t.dens_mean.data(), t.dens_var.data());
    if (rc != SR_OK) { sr_model_destroy(*out); *out = nullptr; }
  }
  return rc;
}
