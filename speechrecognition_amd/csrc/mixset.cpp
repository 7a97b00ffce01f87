// mixset.cpp -- host-side model loading: MIXSET v2 file -> finalised per-density tables -> sr_model.
//
// Mirrors what the reference does once at start-up, MixtureModel::MixtureModel with
// "load-mixtures-from" (sietill/Mixtures.cpp:156-174) -> read() (:748-830) -> finalize() (:374-461,
// calculate_variance :251-275).  The arithmetic below keeps the reference's operation order
// (acc / weight, then  - mean*mean, 1 / var, left-to-right log sum) so that the tables, and with
// them the SR_GMM_EXACT scores, are bit-identical to the reference's private means_/vars_inv_/
// norm_/mean_weights_log_.  Since round 2 the arithmetic runs on the device (em_finalize.hip: srhost::finalize_on_device;
// the logarithms stay on the host's libm); finalize_mixset below is the host-only version of the same, kept as a
// cross-check (SRGPU_HOST_FINALIZE=1).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <cstdlib>

#include "../../include/srgpu.h"
#include "handles.h"
#include "host_util.h"

namespace {

struct Reader {
  FILE* f;
  uint64_t left = 0;  // bytes not yet consumed: every count read from the file is checked against it BEFORE anything is
                      // sized from it, so a corrupt header cannot ask for gigabytes (the reference aborts there)
  explicit Reader(const char* path) : f(fopen(path, "rb")) {
    if (f && fseek(f, 0, SEEK_END) == 0) {
      const long n = ftell(f);
      left = n > 0 ? (uint64_t)n : 0;
      rewind(f);
    }
  }
  ~Reader() { if (f) fclose(f); }
  bool get(void* dst, size_t n) {
    if (n > left || fread(dst, 1, n, f) != n) return false;
    left -= n;
    return true;
  }
  template <typename T> bool get(T* v) { return get(v, sizeof(T)); }
  bool holds(uint64_t count, uint64_t bytes_each) const { return bytes_each == 0 || count <= left / bytes_each; }
};

struct Accumulators {  // one block of read_accumulator (Mixtures.cpp:104-129)
  std::vector<double> sum;     // [n x dim]
  std::vector<double> weight;  // [n]
  uint32_t n = 0;
};

const char* read_block(Reader& r, uint32_t dim, Accumulators* a) {
  if (!r.get(&a->n)) return "Error reading size";
  if (!r.holds(a->n, 4 + 8 * (uint64_t)dim + 8)) return "Error reading features";  // count larger than the file
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

// accumulator-level content of a model: what MixtureModel holds besides the derived tables
struct Mixset {
  uint32_t dim = 0;
  Accumulators mean_acc, var_acc;
  std::vector<std::vector<Density>> mixtures;
};

}  // namespace

namespace srhost {

static const char* parse_mixset(const char* path, uint32_t dim, Mixset* ms) {
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
  Accumulators& mean_acc = ms->mean_acc;
  Accumulators& var_acc = ms->var_acc;
  ms->dim = dim;
  if (const char* e = read_block(r, dim, &mean_acc)) return e;
  if (const char* e = read_block(r, dim, &var_acc)) return e;
  uint32_t n_dens = 0;
  if (!r.get(&n_dens)) return "Error reading density count";
  if (!r.holds(n_dens, 8)) return "Error reading mean_idx";
  std::vector<Density> dens(n_dens);
  for (auto& d : dens) {
    if (!r.get(&d.mean)) return "Error reading mean_idx";
    if (d.mean >= mean_acc.n) return "Invalid mean_idx";
    if (!r.get(&d.var)) return "Error reading var_idx";
    if (d.var >= var_acc.n) return "Invalid var_idx";
  }
  uint32_t n_mix = 0;
  if (!r.get(&n_mix)) return "Error reading mixture count";
  if (!r.holds(n_mix, 4)) return "Error reading density count for mixture";
  std::vector<std::vector<Density>>& mixtures = ms->mixtures;
  mixtures.assign(n_mix, std::vector<Density>());
  for (auto& mix : mixtures) {
    uint32_t nd = 0;
    if (!r.get(&nd)) return "Error reading density count for mixture";
    if (!r.holds(nd, 12)) return "Error reading density idx";
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
  return nullptr;
}

// MixtureModel::finalize (Mixtures.cpp:374-461)
static void finalize_mixset(const Mixset& ms, int pooling, MixsetTables* out) {
  const uint32_t dim = ms.dim;
  const Accumulators& mean_acc = ms.mean_acc;
  const Accumulators& var_acc = ms.var_acc;
  const std::vector<std::vector<Density>>& mixtures = ms.mixtures;

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
  // One mixture at a time, as the reference does.  When no mean or variance row is shared between mixtures (the
  // usual case) the mixtures are independent and are spread over host threads -- every value is computed by the same
  // operations in the same order, so the result does not depend on the thread count; shared rows keep the
  // reference's sequential order (the last writer wins there).
  std::vector<double> mix_totals(mixtures.size(), 0.0);
  auto do_mixture = [&](size_t mi, std::vector<double>& pooled) {
    const std::vector<Density>& mix = mixtures[mi];
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
    mix_totals[mi] = mix_total;
  };
  bool independent = true;
  {
    std::vector<uint32_t> mean_owner(mean_acc.n, 0xFFFFFFFFu), var_owner(var_acc.n, 0xFFFFFFFFu);
    for (size_t mi = 0; mi < mixtures.size() && independent; mi++)
      for (const Density& dn : mixtures[mi]) {
        if ((mean_owner[dn.mean] != 0xFFFFFFFFu && mean_owner[dn.mean] != mi) || (var_owner[dn.var] != 0xFFFFFFFFu && var_owner[dn.var] != mi)) {
          independent = false;
          break;
        }
        mean_owner[dn.mean] = var_owner[dn.var] = (uint32_t)mi;
      }
  }
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t n_threads = (independent && mixtures.size() >= 256) ? std::max(1u, std::min(16u, hw ? hw : 1u)) : 1;
  if (n_threads > 1) {
    parallel_ranges(mixtures.size(), 1, [&](size_t m0, size_t m1) {  // (joined and exception-safe: host_util.h)
      std::vector<double> pooled(D);
      for (size_t mi = m0; mi < m1; mi++) do_mixture(mi, pooled);
    }, n_threads);
  } else {
    std::vector<double> pooled(D);
    for (size_t mi = 0; mi < mixtures.size(); mi++) do_mixture(mi, pooled);
  }
  double total = 0.0;
  for (double mt : mix_totals) total += mt;  // in mixture order, like the reference's running sum
  std::vector<double> pooled(D);
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
}

const char* load_mixset(const char* path, uint32_t dim, int pooling, MixsetTables* out) {
  Mixset ms;
  if (const char* e = parse_mixset(path, dim, &ms)) return e;
  finalize_mixset(ms, pooling, out);
  return nullptr;
}

// in-memory statistics (flattened topology) -> Mixset; returns an error text or nullptr
static const char* mixset_from_arrays(uint32_t dim, uint32_t n_states, const uint32_t* dens_off, uint32_t n_mean, uint32_t n_var,
                                      const uint32_t* dens_mean, const uint32_t* dens_var, const double* mean_acc, const double* mean_w,
                                      const double* var_acc, const double* var_w, Mixset* ms) {
  if (!dens_off || !dens_mean || !dens_var || !mean_acc || !mean_w || !var_acc || !var_w) return "null argument";
  ms->dim = dim;
  ms->mean_acc.n = n_mean; ms->mean_acc.sum.assign(mean_acc, mean_acc + (size_t)n_mean * dim); ms->mean_acc.weight.assign(mean_w, mean_w + n_mean);
  ms->var_acc.n = n_var; ms->var_acc.sum.assign(var_acc, var_acc + (size_t)n_var * dim); ms->var_acc.weight.assign(var_w, var_w + n_var);
  ms->mixtures.assign(n_states, std::vector<Density>());
  for (uint32_t s = 0; s < n_states; s++) {
    if (dens_off[s + 1] < dens_off[s]) return "dens_off must be non-decreasing";
    for (uint32_t c = dens_off[s]; c < dens_off[s + 1]; c++) {
      if (dens_mean[c] >= n_mean || dens_var[c] >= n_var) return "tying index out of range";
      ms->mixtures[s].push_back(Density{dens_mean[c], dens_var[c]});
    }
  }
  return nullptr;
}

// MixtureModel::write (Mixtures.cpp:834-878): rows no density references are dropped and the rest renumbered
// (build_mapping, :83-95); densities are listed in mixture order, so density_idx is a running counter
static const char* write_mixset_file(const char* path, const Mixset& ms) {
  FILE* f = fopen(path, "wb");
  if (!f) return "cannot open output file";
  const uint32_t D = ms.dim;
  std::vector<uint32_t> mean_refs(ms.mean_acc.n, 0), var_refs(ms.var_acc.n, 0), mean_map(ms.mean_acc.n, 0), var_map(ms.var_acc.n, 0);
  for (auto& mix : ms.mixtures)
    for (const Density& dn : mix) { mean_refs[dn.mean]++; var_refs[dn.var]++; }
  uint32_t mean_count = 0, var_count = 0;
  for (uint32_t i = 0; i < ms.mean_acc.n; i++) if (mean_refs[i]) mean_map[i] = mean_count++;
  for (uint32_t i = 0; i < ms.var_acc.n; i++) if (var_refs[i]) var_map[i] = var_count++;
  static const char kMagic[8] = {'M', 'I', 'X', 'S', 'E', 'T', 0, 0};
  const uint32_t version = 2;
  fwrite(kMagic, 1, 8, f); fwrite(&version, 4, 1, f); fwrite(&D, 4, 1, f);
  auto block = [&](const Accumulators& a, const std::vector<uint32_t>& refs, uint32_t count) {
    fwrite(&count, 4, 1, f);
    for (uint32_t i = 0; i < a.n; i++) {
      if (!refs[i]) continue;
      fwrite(&D, 4, 1, f);
      fwrite(a.sum.data() + (size_t)i * D, sizeof(double), D, f);
      fwrite(&a.weight[i], sizeof(double), 1, f);
    }
  };
  block(ms.mean_acc, mean_refs, mean_count);
  block(ms.var_acc, var_refs, var_count);
  uint32_t density_count = 0;
  for (auto& mix : ms.mixtures) density_count += (uint32_t)mix.size();
  fwrite(&density_count, 4, 1, f);
  for (auto& mix : ms.mixtures)
    for (const Density& dn : mix) { fwrite(&mean_map[dn.mean], 4, 1, f); fwrite(&var_map[dn.var], 4, 1, f); }
  const uint32_t mixture_count = (uint32_t)ms.mixtures.size();
  fwrite(&mixture_count, 4, 1, f);
  uint32_t running = 0;
  for (auto& mix : ms.mixtures) {
    const uint32_t nd = (uint32_t)mix.size();
    fwrite(&nd, 4, 1, f);
    for (const Density& dn : mix) { fwrite(&running, 4, 1, f); fwrite(&ms.mean_acc.weight[dn.mean], sizeof(double), 1, f); running++; }
  }
  const bool ok = !ferror(f);
  fclose(f);
  return ok ? nullptr : "write error";
}

}  // namespace srhost

using srhost::guarded;

// flat topology + accumulators of a parsed file, for the device-side finalize
static int mixset_to_model(const Mixset& ms, int pooling, int max_approx, int device, sr_model** out) {
  std::vector<uint32_t> dens_off(1, 0u), dens_mean, dens_var;
  for (auto& mix : ms.mixtures) {
    for (const Density& dn : mix) { dens_mean.push_back(dn.mean); dens_var.push_back(dn.var); }
    dens_off.push_back((uint32_t)dens_mean.size());
  }
  if (dens_mean.empty()) { dens_mean.push_back(0); dens_var.push_back(0); }  // (non-null pointers for an empty model)
  return srhost::finalize_on_device(device, ms.dim, (uint32_t)ms.mixtures.size(), dens_off.data(), ms.mean_acc.n, ms.var_acc.n,
                                    dens_mean.data(), dens_var.data(), ms.mean_acc.sum.data(), ms.mean_acc.weight.data(),
                                    ms.var_acc.sum.data(), ms.var_acc.weight.data(), pooling, max_approx, out);
}
static bool host_finalize_requested() {
  const char* e = getenv("SRGPU_HOST_FINALIZE");
  return e && atoi(e) != 0;
}

static int tables_to_model(const srhost::MixsetTables& t, uint32_t dim, int max_approx, int device, sr_model** out) {
  int rc = sr_model_create(device, dim, (uint32_t)t.dens_off.size() - 1, t.dens_off.data(), t.means.data(), t.inv_vars.data(),
                           t.norm.data(), t.logw.data(), max_approx, out);
  if (rc == SR_OK && !t.dens_mean.empty()) {
    rc = sr_model_set_tying(*out, t.n_mean, t.n_var, t.dens_mean.data(), t.dens_var.data());
    if (rc != SR_OK) { sr_model_destroy(*out); *out = nullptr; }
  }
  return rc;
}

extern "C" SR_API int sr_model_create_from_statistics(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off,
                                                      uint32_t n_mean, uint32_t n_var, const uint32_t* dens_mean,
                                                      const uint32_t* dens_var, const double* mean_acc, const double* mean_w,
                                                      const double* var_acc, const double* var_w, int pooling, int max_approx,
                                                      sr_model** out) {
  return guarded(__func__, [&]() -> int {
  if (!out) return srhost::set_error(SR_EINVAL, "out is null");
  if (pooling < 0 || pooling > 2) return srhost::set_error(SR_EINVAL, "pooling must be 0 (global), 1 (mixture) or 2 (none)");
  if (!host_finalize_requested())
    return srhost::finalize_on_device(device, dim, n_states, dens_off, n_mean, n_var, dens_mean, dens_var, mean_acc, mean_w, var_acc,
                                      var_w, pooling, max_approx, out);
  Mixset ms;
  if (const char* e = srhost::mixset_from_arrays(dim, n_states, dens_off, n_mean, n_var, dens_mean, dens_var, mean_acc, mean_w, var_acc, var_w, &ms))
    return srhost::set_error(SR_EINVAL, e);
  srhost::MixsetTables t;
  srhost::finalize_mixset(ms, pooling, &t);
  return tables_to_model(t, dim, max_approx, device, out);
  });
}

extern "C" SR_API int sr_mixset_write(const char* path, uint32_t dim, uint32_t n_states, const uint32_t* dens_off, uint32_t n_mean,
                                      uint32_t n_var, const uint32_t* dens_mean, const uint32_t* dens_var, const double* mean_acc,
                                      const double* mean_w, const double* var_acc, const double* var_w) {
  return guarded(__func__, [&]() -> int {
  if (!path) return srhost::set_error(SR_EINVAL, "path is null");
  Mixset ms;
  if (const char* e = srhost::mixset_from_arrays(dim, n_states, dens_off, n_mean, n_var, dens_mean, dens_var, mean_acc, mean_w, var_acc, var_w, &ms))
    return srhost::set_error(SR_EINVAL, e);
  if (const char* e = srhost::write_mixset_file(path, ms)) return srhost::set_error(SR_EINVAL, e);
  return SR_OK;
  });
}

extern "C" SR_API int sr_model_load_mixset(const char* path, uint32_t dim, int pooling, int max_approx, int device,
                                           sr_model** out) {
  return guarded(__func__, [&]() -> int {
  if (!path || !out) return srhost::set_error(SR_EINVAL, "null argument");
  if (pooling < 0 || pooling > 2) return srhost::set_error(SR_EINVAL, "pooling must be 0 (global), 1 (mixture) or 2 (none)");
  if (!host_finalize_requested()) {
    Mixset ms;
    if (const char* e = srhost::parse_mixset(path, dim, &ms)) return srhost::set_error(SR_EINVAL, e);
    if (ms.mixtures.empty()) return srhost::set_error(SR_EINVAL, "dim and n_states must be positive");
    return mixset_to_model(ms, pooling, max_approx, device, out);
  }
  srhost::MixsetTables t;
  if (const char* e = srhost::load_mixset(path, dim, pooling, &t)) return srhost::set_error(SR_EINVAL, e);
  return tables_to_model(t, dim, max_approx, device, out);
  });
}
