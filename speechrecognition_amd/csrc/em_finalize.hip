// em_finalize.hip -- MixtureModel::finalize (sietill/Mixtures.cpp:374-461, calculate_variance :251-275) on the device: the
// model update that closes an EM iteration (sr_model_create_from_statistics).  The accumulators arrive from
// sr_accumulate_corpus (possibly all-reduced over ranks) as host arrays; the new model's tables are built where the scoring
// kernels read them, in HBM, and the host keeps no copy of them.
//
// Arithmetic, in the reference's order and without FMA (-ffp-contract=off; v_div sequences are correctly rounded):
//   means[i][d]  = mean_acc[i][d] / mean_w[i]                                               (:390-393)
//   v            = var_acc[j][d] / var_w[j];  vars[j][d] = v - mu[d] * mu[d];  vars_inv = 1 / vars    (calculate_variance)
//       mu = NO_POOLING:      the mean of the LAST density (mixture order) that references variance row j (:396-398 runs
//                             once per referencing density; the last one stays)
//            MIXTURE_POOLING: sum of the mixture's mean accumulators (left to right from 0.0) / the mixture's observations,
//                             into the variance row of the mixture's FIRST density (:408-427); last mixture wins
//            GLOBAL_POOLING:  the same sum over the whole model / all observations, into row 0 only (:431-450)
//       a row nobody writes keeps read()'s zero-initialised vars_ / vars_inv_ / norm_ (:776-778)
// What stays on the host: every log().  norm[j] = (D log 2 pi + log v_0 + log v_1 + ...) / 2 and logw[i] = log(w_i / mixture
// observations) go through libm there, because the tables must carry the reference's bits and the device's log (1 ulp, another
// algorithm) does not reproduce glibc's; the variances come back for it (n_var x D doubles) and 2 x C doubles go up.
// Observation sums per mixture / model are sequential host sums in mixture order, like the reference's running totals.
#include <hip/hip_runtime.h>

#include <cmath>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "handles.h"

#pragma clang fp contract(off)

using srhost::fail;

namespace {

constexpr uint32_t kNoWriter = 0xFFFFFFFFu;

__global__ void fin_means_kernel(const double* acc, const double* w, uint32_t n_rows, uint32_t D, double* means) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)n_rows * D) return;
  means[i] = acc[i] / w[i / D];
}

// pooled[g][d] = (sum over the densities c of group g, in order, of mean_acc[dens_mean[c]][d]) / total[g]
// one thread per (group, dimension): the sum is sequential, like the reference's std::transform chain
__global__ void fin_pooled_kernel(const double* mean_acc, const uint32_t* dens_mean, const uint32_t* grp_off, const double* grp_total,
                                  uint32_t n_groups, uint32_t D, double* pooled) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)n_groups * D) return;
  const uint32_t g = (uint32_t)(i / D), d = (uint32_t)(i % D);
  double s = 0.0;
  for (uint32_t c = grp_off[g]; c < grp_off[g + 1]; c++) s = s + mean_acc[(uint64_t)dens_mean[c] * D + d];
  pooled[i] = s / grp_total[g];
}

// calculate_variance for every variance row: src[j] = row of `mu` to take (kNoWriter: the row is never finalised)
__global__ void fin_vars_kernel(const double* var_acc, const double* var_w, const double* mu, const uint32_t* src, uint32_t n_var,
                                uint32_t D, double* vars, double* ivars) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)n_var * D) return;
  const uint32_t j = (uint32_t)(i / D), d = (uint32_t)(i % D), r = src[j];
  if (r == kNoWriter) { vars[i] = 0.0; ivars[i] = 0.0; return; }
  double v = var_acc[i] / var_w[j];
  const double m = mu[(uint64_t)r * D + d];
  v = v - m * m;
  vars[i] = v;
  ivars[i] = 1 / v;
}

// per-density tables in mixture order: what the scoring kernels index
__global__ void fin_expand_kernel(const double* means, const double* ivars, const uint32_t* dens_mean, const uint32_t* dens_var,
                                  uint64_t n_dens, uint32_t D, double* means_e, double* ivars_e) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_dens * D) return;
  const uint64_t c = i / D, d = i % D;
  means_e[i] = means[(uint64_t)dens_mean[c] * D + d];
  ivars_e[i] = ivars[(uint64_t)dens_var[c] * D + d];
}

// One pinned host buffer for the process, grown on demand and kept: hipHostMalloc costs milliseconds, an EM loop calls
// finalize once per iteration.  A lease holds the buffer's mutex (finalize calls of several device threads take turns).
struct PinnedScratch {
  std::mutex mu;
  void* p = nullptr;
  size_t bytes = 0;
  static PinnedScratch& instance() { static PinnedScratch s; return s; }
  struct Lease {
    PinnedScratch& s;
    void* ptr = nullptr;
    Lease(PinnedScratch& s_, size_t need) : s(s_) {
      s.mu.lock();
      if (need > s.bytes) {
        if (s.p) (void)hipHostFree(s.p);
        s.p = nullptr; s.bytes = 0;
        void* q = nullptr;
        if (hipHostMalloc(&q, need, hipHostMallocPortable) == hipSuccess) { s.p = q; s.bytes = need; }
        else (void)hipGetLastError();
      }
      ptr = s.p;
    }
    ~Lease() { s.mu.unlock(); }
  };
};

template <typename T>
hipError_t to_device(DevBuf<T>& b, const T* src, size_t n) { return b.upload(src, n); }

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

namespace srhost {

// acc_on_device: mean_acc / var_acc are DEVICE pointers (the accumulators sr_accumulate_corpus left in the corpus handle);
// the weights are host arrays either way (the observation totals and log weights are host work)
static int finalize_core(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off, uint32_t n_mean, uint32_t n_var,
                         const uint32_t* dens_mean, const uint32_t* dens_var, const double* mean_acc, const double* mean_w,
                         const double* var_acc, const double* var_w, bool acc_on_device, int pooling, int max_approx, sr_model** out) {
  *out = nullptr;
  static const bool fin_timing = getenv("SRGPU_FIN_TIMING") != nullptr;  // phase times to stderr (diagnostic)
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now_ms();
  double t_last = t_begin;
  auto lap = [&](const char* what) {
    if (!fin_timing) return;
    const double t = now_ms();
    fprintf(stderr, "[finalize] %-28s %7.2f ms\n", what, t - t_last);
    t_last = t;
  };
  if (!dens_off || !dens_mean || !dens_var || !mean_acc || !mean_w || !var_acc || !var_w) return fail(SR_EINVAL, "null argument");
  for (uint32_t s = 0; s < n_states; s++)
    if (dens_off[s + 1] < dens_off[s]) return fail(SR_EINVAL, "dens_off must be non-decreasing");
  const uint64_t C = dens_off[n_states];
  for (uint64_t c = 0; c < C; c++)
    if (dens_mean[c] >= n_mean || dens_var[c] >= n_var) return fail(SR_EINVAL, "tying index out of range");
  sr_model* m = nullptr;
  int rc = model_shell(device, dim, n_states, dens_off, max_approx, &m);
  if (rc != SR_OK) return rc;
  std::unique_ptr<sr_model, int (*)(sr_model*)> own(m, sr_model_destroy);
  const uint32_t D = dim;
  hipStream_t st = m->s_gmm;
  lap("checks + model shell");

  // ---- host: observation totals, last writers (integer work + C sequential additions) ------------------------------
  std::vector<double> mix_total(n_states, 0.0), logw_e(C, 0.0);
  double total = 0.0;
  std::vector<uint32_t> var_src(n_var, kNoWriter), mean_last(n_mean, kNoWriter);
  for (uint32_t s = 0; s < n_states; s++) {
    double t = 0.0;
    for (uint32_t c = dens_off[s]; c < dens_off[s + 1]; c++) {
      t += mean_w[dens_mean[c]];
      if (pooling == SRHOST_POOL_NONE) var_src[dens_var[c]] = dens_mean[c];  // the last referencing density stays
      mean_last[dens_mean[c]] = s;                                            // ... and the last mixture's log weight
    }
    mix_total[s] = t;
    if (pooling == SRHOST_POOL_MIXTURE && dens_off[s + 1] > dens_off[s]) var_src[dens_var[dens_off[s]]] = s;  // row of the pooled table
    total += t;
  }
  if (pooling == SRHOST_POOL_GLOBAL && n_var > 0) var_src[0] = 0;
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t n_threads = C >= 4096 ? std::max(1u, std::min(16u, hw ? hw : 1u)) : 1;
  auto spread = [&](size_t n, auto&& fn) {  // fn(i0, i1) over [0, n) on n_threads host threads (joined and exception-safe: host_util.h)
    if (n_threads <= 1 || n < 1024) { fn((size_t)0, n); return; }
    srhost::parallel_ranges(n, 1, fn, n_threads);
  };
  // log weights per density: log(w / observations of the LAST mixture that references the mean row) (:401-405)
  spread(C, [&](size_t c0, size_t c1) {
    for (size_t c = c0; c < c1; c++) logw_e[c] = log(mean_w[dens_mean[c]] / mix_total[mean_last[dens_mean[c]]]);
  });

  lap("host totals + log weights");
  // ---- device: divisions, variances, per-density expansion -----------------------------------------------------------
  DevBuf<double> d_macc_own, d_mw, d_vacc_own, d_vw, d_means, d_vars, d_ivars, d_pooled, d_total;
  DevBuf<uint32_t> d_src, d_goff;
  hipError_t e;
  if (!acc_on_device &&
      ((e = to_device(d_macc_own, mean_acc, (size_t)n_mean * D)) != hipSuccess || (e = to_device(d_vacc_own, var_acc, (size_t)n_var * D)) != hipSuccess))
    return fail(SR_EHIP, "finalize upload: %s", hipGetErrorString(e));
  struct { const double* p; } d_macc = {acc_on_device ? mean_acc : d_macc_own.p}, d_vacc = {acc_on_device ? var_acc : d_vacc_own.p};
  if ((e = to_device(d_mw, mean_w, n_mean)) != hipSuccess || (e = to_device(d_vw, var_w, n_var)) != hipSuccess ||
      (e = to_device(d_src, var_src.data(), n_var)) != hipSuccess || (e = m->dens_mean.upload(dens_mean, C)) != hipSuccess ||
      (e = m->dens_var.upload(dens_var, C)) != hipSuccess || (e = d_means.ensure((size_t)n_mean * D)) != hipSuccess ||
      (e = d_vars.ensure((size_t)n_var * D)) != hipSuccess || (e = d_ivars.ensure((size_t)n_var * D)) != hipSuccess ||
      (e = m->means.ensure(C * D)) != hipSuccess || (e = m->inv_vars.ensure(C * D)) != hipSuccess ||
      (e = m->norm.ensure(C)) != hipSuccess || (e = m->logw.ensure(C)) != hipSuccess)
    return fail(SR_EHIP, "finalize upload: %s", hipGetErrorString(e));
  m->n_mean = n_mean; m->n_var = n_var;
  m->h_dens_mean.assign(dens_mean, dens_mean + C);
  m->h_dens_var.assign(dens_var, dens_var + C);
  if ((uint64_t)n_mean * D) hipLaunchKernelGGL(fin_means_kernel, grid_for((uint64_t)n_mean * D), dim3(256), 0, st, d_macc.p, d_mw.p, n_mean, D, d_means.p);
  const double* mu = d_means.p;
  if (pooling == SRHOST_POOL_MIXTURE) {
    if ((e = to_device(d_goff, dens_off, (size_t)n_states + 1)) != hipSuccess || (e = to_device(d_total, mix_total.data(), n_states)) != hipSuccess ||
        (e = d_pooled.ensure((size_t)n_states * D)) != hipSuccess)
      return fail(SR_EHIP, "finalize upload: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(fin_pooled_kernel, grid_for((uint64_t)n_states * D), dim3(256), 0, st, d_macc.p, m->dens_mean.p, d_goff.p, d_total.p, n_states, D, d_pooled.p);
    mu = d_pooled.p;
  } else if (pooling == SRHOST_POOL_GLOBAL) {
    const uint32_t goff[2] = {0u, (uint32_t)C};
    if ((e = to_device(d_goff, goff, 2)) != hipSuccess || (e = to_device(d_total, &total, 1)) != hipSuccess || (e = d_pooled.ensure(D)) != hipSuccess)
      return fail(SR_EHIP, "finalize upload: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(fin_pooled_kernel, grid_for(D), dim3(256), 0, st, d_macc.p, m->dens_mean.p, d_goff.p, d_total.p, 1u, D, d_pooled.p);
    mu = d_pooled.p;
  }
  if ((uint64_t)n_var * D) hipLaunchKernelGGL(fin_vars_kernel, grid_for((uint64_t)n_var * D), dim3(256), 0, st, d_vacc.p, d_vw.p, mu, d_src.p, n_var, D, d_vars.p, d_ivars.p);
  if (C) hipLaunchKernelGGL(fin_expand_kernel, grid_for(C * D), dim3(256), 0, st, d_means.p, d_ivars.p, m->dens_mean.p, m->dens_var.p, C, D, m->means.p, m->inv_vars.p);
  HIP_TRY(hipGetLastError());
  lap("uploads + kernel launches");

  // ---- host: the logarithms (libm: the reference's bits), on the variances the device computed.  The variances come back
  // in pieces through a pinned buffer kept for the process (pageable memory, zero-filled first, made this the longest part of
  // an EM iteration), and every host thread starts on a piece as soon as its copy has landed ----------------------------
  std::vector<double> norm_row(n_var, 0.0);
  if (n_var) {
    PinnedScratch::Lease lease(PinnedScratch::instance(), (size_t)n_var * D * sizeof(double));
    const double* vars = static_cast<const double*>(lease.ptr);
    constexpr int kPieces = 8;
    hipEvent_t ev[kPieces];
    size_t row_lo[kPieces + 1];
    for (int k = 0; k <= kPieces; k++) row_lo[k] = (size_t)n_var * k / kPieces;
    bool pinned = vars != nullptr;
    std::vector<double> pageable;
    if (!pinned) {  // no pinned memory to be had: one blocking copy into ordinary memory
      pageable.resize((size_t)n_var * D);
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipMemcpy(pageable.data(), d_vars.p, pageable.size() * sizeof(double), hipMemcpyDeviceToHost));
      vars = pageable.data();
    } else {
      for (int k = 0; k < kPieces; k++) {
        HIP_TRY(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        const size_t off = row_lo[k] * D, cnt = (row_lo[k + 1] - row_lo[k]) * D;
        if (cnt) HIP_TRY(hipMemcpyAsync(const_cast<double*>(vars) + off, d_vars.p + off, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(ev[k], st));
      }
    }
    std::atomic<int> failed{0};
    auto rows = [&](size_t t, size_t nt) {
      for (int k = 0; k < kPieces; k++) {
        if (pinned && hipEventSynchronize(ev[k]) != hipSuccess) { failed = 1; return; }
        const size_t n = row_lo[k + 1] - row_lo[k], j0 = row_lo[k] + n * t / nt, j1 = row_lo[k] + n * (t + 1) / nt;
        for (size_t j = j0; j < j1; j++) {
          if (var_src[j] == kNoWriter) continue;  // norm_ stays 0
          double acc = D * log(2 * M_PI);
          for (uint32_t d = 0; d < D; d++) acc = acc + log(vars[j * D + d]);
          norm_row[j] = acc / 2;
        }
      }
    };
    if (n_threads <= 1 || n_var < 1024) {
      rows(0, 1);
    } else {
      srhost::ThreadGroup pool(n_threads - 1);
      for (size_t t = 0; t + 1 < n_threads; t++) pool.run([&rows, t, n_threads]() { rows(t, n_threads); });
      pool.run_here([&]() { rows(n_threads - 1, n_threads); });
      pool.wait();
    }
    if (pinned)
      for (int k = 0; k < kPieces; k++) (void)hipEventDestroy(ev[k]);
    if (failed) return fail(SR_EHIP, "finalize: waiting for the variances failed");
  } else {
    HIP_TRY(hipStreamSynchronize(st));
  }
  lap("variances back + logarithms");
  std::vector<double> norm_e(C);
  for (uint64_t c = 0; c < C; c++) norm_e[c] = norm_row[dens_var[c]];
  if (C) {
    HIP_TRY(hipMemcpy(m->norm.p, norm_e.data(), C * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->logw.p, logw_e.data(), C * sizeof(double), hipMemcpyHostToDevice));
  }
  lap("norm gather + upload");
  *out = own.release();
  return SR_OK;
}

int finalize_on_device(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off, uint32_t n_mean, uint32_t n_var,
                       const uint32_t* dens_mean, const uint32_t* dens_var, const double* mean_acc, const double* mean_w,
                       const double* var_acc, const double* var_w, int pooling, int max_approx, sr_model** out) {
  return finalize_core(device, dim, n_states, dens_off, n_mean, n_var, dens_mean, dens_var, mean_acc, mean_w, var_acc, var_w, false,
                       pooling, max_approx, out);
}

// the statistics sr_accumulate_corpus left on the device (corpus handle) -> new model, nothing but the weights crosses PCIe
int finalize_accumulated(sr_model* m, sr_corpus* c, int pooling, int max_approx, sr_model** out) {
  *out = nullptr;
  if (!c->acc_valid || c->acc_n_mean != m->n_mean || c->acc_n_var != m->n_var)
    return fail(SR_EINVAL, "the corpus holds no statistics of this model: call sr_accumulate_corpus first");
  std::vector<double> mw(std::max(1u, m->n_mean)), vw(std::max(1u, m->n_var));
  HIP_TRY(hipMemcpy(mw.data(), c->w_mean.p, sizeof(double) * m->n_mean, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(vw.data(), c->w_var.p, sizeof(double) * m->n_var, hipMemcpyDeviceToHost));
  if (m->h_dens_mean.empty()) return fail(SR_EINVAL, "model without densities");
  return finalize_core(m->device, m->dim, m->n_states, m->h_dens_off.data(), m->n_mean, m->n_var, m->h_dens_mean.data(),
                       m->h_dens_var.data(), c->acc_mean.p, mw.data(), c->acc_var.p, vw.data(), true, pooling, max_approx, out);
}

}  // namespace srhost
