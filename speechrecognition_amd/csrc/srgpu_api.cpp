// srgpu_api.cpp -- C ABI of libsrgpu.so (include/srgpu.h): handles, device memory, model packing,
// the chunked score -> search pipeline on two HIP streams, and event-based kernel timing.
//
// Nothing here computes scores or paths on the host: every sr_score_* / sr_recognize_* / sr_align_*
// call runs the HIP kernels of gmm_mfma.hip / gmm_exact.hip / viterbi_decode.hip / viterbi_align.hip
// and fails loudly (SR_EHIP / SR_ENODEV) when no gfx950 device is usable.  There is no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/srgpu.h"
#include "host_util.h"
#include "kernels.h"
#include "traceback.h"

using namespace srgpu;
using srhost::guarded;

#include "handles.h"

using srhost::fail;

namespace {
thread_local char g_err[512] = "";
}  // namespace
namespace srhost {
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }
}  // namespace srhost

namespace {

// ---- model packing for the MFMA kernel ---------------------------------------------------------
// States are sorted by density count (stable, descending) and chunked into groups of four; group q
// owns ceil(max_count/4) blocks of 16 model rows; row r of block j holds density 4*j + (r>>2) of the
// group's state slot (r&3).  Fragment order: apack[block][kstep][lane] = A[row = lane&15][k = 4*kstep + (lane>>4)].
int pack_model(sr_model* m, const uint32_t* dens_off, const double* means, const double* inv_vars,
               const double* norm, const double* logw) {
  const uint32_t S = m->n_states, D = m->dim;
  const int KS = m->ksteps;
  std::vector<uint32_t> order(S);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    return (dens_off[x + 1] - dens_off[x]) > (dens_off[y + 1] - dens_off[y]);
  });
  const uint32_t n_groups = (S + 3) / 4;
  std::vector<uint32_t> grp_state(4 * (size_t)n_groups, 0xFFFFFFFFu), first_block(n_groups + 1, 0);
  for (uint32_t q = 0; q < n_groups; q++) {
    uint32_t mx = 0;
    for (uint32_t g = 0; g < 4; g++) {
      const uint32_t i = 4 * q + g;
      if (i < S) {
        grp_state[4 * q + g] = order[i];
        mx = std::max(mx, dens_off[order[i] + 1] - dens_off[order[i]]);
      }
    }
    first_block[q + 1] = first_block[q] + std::max(1u, (mx + 3) / 4);
  }
  const uint32_t n_blocks = first_block[n_groups];
  const size_t blk_doubles = (size_t)KS * 64;
  std::vector<double> apack(blk_doubles * n_blocks, 0.0);
  std::vector<uint32_t> meta(n_blocks);
  const double inf = std::numeric_limits<double>::infinity();
  for (uint32_t q = 0; q < n_groups; q++) {
    for (uint32_t b = first_block[q]; b < first_block[q + 1]; b++) {
      meta[b] = (q << 1) | (b + 1 == first_block[q + 1] ? 1u : 0u);
      const uint32_t j = b - first_block[q];
      for (uint32_t r = 0; r < 16; r++) {
        const uint32_t g = r & 3, sub = r >> 2;
        const uint32_t st = grp_state[4 * q + g];
        const uint32_t i = 4 * j + sub;
        const bool real = st != 0xFFFFFFFFu && i < dens_off[st + 1] - dens_off[st];
        const size_t c = real ? (size_t)dens_off[st] + i : 0;
        double konst = inf;  // padding row: +inf never wins the min and adds exp(-inf) = 0 to a sum
        if (real) {
          double q2 = 0.0;
          for (uint32_t d = 0; d < D; d++) q2 += means[c * D + d] * means[c * D + d] * inv_vars[c * D + d];
          konst = norm[c] - logw[c] + 0.5 * q2;
        }
        for (int ks = 0; ks < KS; ks++) {
          for (uint32_t kk = 0; kk < 4; kk++) {
            const uint32_t k = 4 * ks + kk;
            double v = 0.0;
            if (k < 2 * D) {
              if (real) {
                const uint32_t d = k >> 1;
                v = (k & 1) ? -means[c * D + d] * inv_vars[c * D + d] : 0.5 * inv_vars[c * D + d];
              }
            } else if (k == 2 * D) {
              v = konst;
            }
            apack[(size_t)b * blk_doubles + (size_t)ks * 64 + kk * 16 + r] = v;
          }
        }
      }
    }
  }
  m->n_groups = n_groups;
  m->n_blocks = n_blocks;
  m->group_first_block = first_block;
  HIP_TRY(m->apack.upload(apack.data(), apack.size()));
  HIP_TRY(m->blk_meta.upload(meta.data(), meta.size()));
  HIP_TRY(m->grp_state.upload(grp_state.data(), grp_state.size()));
  return SR_OK;
}

// ---- model packing for the fp16 prefilter ---------------------------------------------------------------------
// Groups of four (pseudo-)states in natural order, each padded to 8 blocks of 16 rows (32 density slots per state);
// row r of block j = density 4*j + (r & 3) of state slot (r >> 2).  Coefficients are scaled by a power of two and
// rounded once to fp16; fragment order [group][block][k-step of 32][lane][8]: lane l holds row l & 15,
// k = 32*ks + 8*(l >> 4) + j -- the A operand of v_mfma_f32_16x16x32_f16.
int pack_prefilter(sr_model* m, const uint32_t* dens_off, const double* means, const double* inv_vars,
                   const double* norm, const double* logw) {
  const uint32_t S = m->n_states, D = m->dim;
  uint32_t mx = 0;
  for (uint32_t s = 0; s < S; s++) mx = std::max(mx, dens_off[s + 1] - dens_off[s]);
  m->max_dens = std::max(1u, mx);
  m->pf_ks32 = 0;
  const uint32_t DP = gmm_refine_padded_dim(D);  // the odd dimension the refinement runs in (pairs | zeros | odd tail)
  // not eligible: callers get the exact kernel (> 128 densities per mixture need all eight panels of a state in one workgroup's LDS)
  if (!m->max_approx || mx > 256 || (mx > 128 && DP > 39) || 2 * D + 3 > 128 || DP == 0) return SR_OK;
  const int KS = (int)((2 * D + 3 + 31) / 32);
  {
    bool preserved = false;
    HIP_TRY(probe_fp16_denormals(m->s_gmm, &preserved));
    if (!preserved) return SR_OK;  // the bound of gmm_prefilter.hip does not hold with flushed subnormals: exact kernel
    bool accumulates = false;
    HIP_TRY(probe_fp16_accumulation(m->s_gmm, KS <= 3 ? 3 : 4, &accumulates, nullptr));  // the chain length this model runs
    if (!accumulates) return SR_OK;  // ... nor with an accumulator outside its model
  }
  m->pf_dp = DP;
  // A mixture of more than 32 densities is cut into Cs = 2 or 4 chunks of 32: the kernels see S*Cs pseudo-states
  // (ps = st*Cs + chunk), the prefilter takes the minimum across a state's chunks, the refinement folds them.
  // (129 .. 256 densities: Cs = 8 = two halves of four chunks; the fp16 pass sees each half as a four-chunk state of its own)
  const uint32_t Cs = mx <= 32 ? 1u : mx <= 64 ? 2u : mx <= 128 ? 4u : 8u;
  const uint32_t PS = S * Cs;
  m->pf_chunks = Cs;
  m->pf_pstates = PS;
  auto ps_count = [&](uint32_t ps) -> uint32_t {  // densities of pseudo-state ps
    const uint32_t n = dens_off[ps / Cs + 1] - dens_off[ps / Cs], lo = 32u * (ps % Cs);
    return n > lo ? std::min(32u, n - lo) : 0u;
  };
  {
    std::vector<uint32_t> cnt(PS);
    for (uint32_t ps = 0; ps < PS; ps++) cnt[ps] = ps_count(ps);
    HIP_TRY(m->pf_ndens.upload(cnt.data(), cnt.size()));
  }
  // FP64 planes for the refinement: [pseudo-state][mu_0 | 1/var_0 | ... | norm | logw][density slot]; 1 KB of slack
  // for the LDS-DMA's last piece
  {
    const uint32_t NS = (uint32_t)gmm_refine_slots(std::min(32u, m->max_dens), DP), planes = 2 * DP + 2;
    // padded plane order (gmm_refine_kernel): dimensions 0 .. D - D % 2 - 1 in place, zero planes, an odd D's last dimension in slot
    // DP - 1; a zero plane pair (mean 0, inverse variance 0) contributes +0.0 to the sums
    const uint32_t pairs = D - (D & 1u);
    auto slot_of = [&](uint32_t d) -> uint32_t { return d < pairs ? d : DP - 1u; };
    m->pf_slots = NS;
    // (the packings are built on first use of their kernel: on 16 host threads, a new model's first prefilter call was 170 ms)
    const size_t n_rows_d = (size_t)PS * planes * NS + 128;
    std::unique_ptr<double[]> rows_own(new double[n_rows_d]);
    double* const rows_p = rows_own.get();
    std::fill(rows_p + (size_t)PS * planes * NS, rows_p + n_rows_d, 0.0);
    srhost::parallel_ranges(PS, 64, [&](size_t ps0, size_t ps1) {
    for (uint32_t ps = (uint32_t)ps0; ps < (uint32_t)ps1; ps++) {
      double* r = rows_p + (size_t)ps * planes * NS;
      std::fill(r, r + (size_t)planes * NS, 0.0);
      for (uint32_t i = 0; i < ps_count(ps); i++) {
        const size_t c = (size_t)dens_off[ps / Cs] + 32u * (ps % Cs) + i;
        // density i sits in slot i (round 1 rotated a workgroup's states against each other for a phase in which a wave
        // instruction mixed states; since round 2 every wave instruction evaluates candidates of ONE state, so lanes on
        // different densities hit different bank pairs and lanes on the same density share a broadcast read as it is)
        const uint32_t sl = i;
        for (uint32_t d = 0; d < D; d++) { r[(2 * slot_of(d)) * NS + sl] = means[c * D + d]; r[(2 * slot_of(d) + 1) * NS + sl] = inv_vars[c * D + d]; }
        r[(2 * DP) * NS + sl] = norm[c];
        r[(2 * DP + 1) * NS + sl] = logw[c];
      }
    }
    });
    HIP_TRY(m->pf_rows.upload(rows_p, n_rows_d));
  }
  const uint32_t n_groups = (PS + 3) / 4;
  const float finf = std::numeric_limits<float>::infinity();
  // per density: coefficients a = [1/(2 var); -mu/var] and the constant
  auto coeffs = [&](size_t c, std::vector<double>& arow) -> double {
    double q2 = 0.0;
    for (uint32_t d = 0; d < D; d++) {
      const double mu = means[c * D + d], iv = inv_vars[c * D + d];
      arow[2 * d] = 0.5 * iv;
      arow[2 * d + 1] = -mu * iv;
      q2 += mu * mu * iv;
    }
    return norm[c] - logw[c] + 0.5 * q2;
  };
  // one power-of-two scale for the whole model, the largest finite |coefficient| or |constant| -> [2^13, 2^14)
  double sA = 1.0;
  {
    std::mutex mu_big;
    double big = 0.0;
    srhost::parallel_ranges((size_t)m->n_dens, 4096, [&](size_t c0, size_t c1) {
      std::vector<double> arow(32 * (size_t)KS);
      double mine = 0.0;
      for (size_t c = c0; c < c1; c++) {
        const double konst = coeffs(c, arow);
        if (std::isfinite(konst)) mine = std::max(mine, std::fabs(konst));
        for (uint32_t k = 0; k < 2 * D; k++)
          if (std::isfinite(arow[k])) mine = std::max(mine, std::fabs(arow[k]));
      }
      std::lock_guard<std::mutex> lock(mu_big);
      big = std::max(big, mine);
    });
    if (big > 0.0) sA = std::ldexp(1.0, 13 - std::ilogb(big));
  }
  const size_t blk_bytes = (size_t)KS * 1024;
  std::vector<uint16_t> ap((size_t)n_groups * 8 * blk_bytes / 2, 0);
  std::vector<float> anorm(8 * (size_t)n_groups, 0.0f);  // (sA |a|, sA |konst|) per state slot
  auto half_bits = [](double v) { const _Float16 h = (_Float16)(float)v; uint16_t u; memcpy(&u, &h, 2); return u; };
  auto half_value = [](uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (double)(float)h; };
  srhost::parallel_ranges(n_groups, 16, [&](size_t q0, size_t q1) {
  std::vector<double> arow(32 * (size_t)KS);
  for (uint32_t q = (uint32_t)q0; q < (uint32_t)q1; q++) {
    for (uint32_t j = 0; j < 8; j++) {
      const size_t b = (size_t)q * 8 + j;
      for (uint32_t r = 0; r < 16; r++) {
        const uint32_t g = r >> 2, ps = 4 * q + g, i = 4 * j + (r & 3);
        const bool real = ps < PS && i < ps_count(ps);
        const uint32_t st = ps < PS ? ps / Cs : 0;
        std::fill(arow.begin(), arow.end(), 0.0);
        double konst = (double)finf;  // padding slot: never below a real score, masked off again by the refinement
        if (real) {
          konst = coeffs((size_t)dens_off[st] + 32u * (ps % Cs) + i, arow) * sA;
          double n2 = 0.0;
          for (uint32_t k = 0; k < 2 * D; k++) { arow[k] *= sA; n2 += arow[k] * arow[k]; }
          const float na = std::nextafter((float)(std::sqrt(n2) * (1.0 + 1e-6)), finf);
          const float nk = std::nextafter((float)(std::fabs(konst) * (1.0 + 1e-6)), finf);
          // NaN sticks: everything of that state then stays a candidate
          if (!(na <= anorm[2 * (4 * q + g)])) anorm[2 * (4 * q + g)] = na;
          if (!(nk <= anorm[2 * (4 * q + g) + 1])) anorm[2 * (4 * q + g) + 1] = nk;
        }
        // konst = c1 + c2 + c3 (3 x 11 bits), multiplied by 1 in three spare k slots: exact products
        double rest = konst;
        for (uint32_t t = 0; t < 3; t++) {
          const double c = (real || t == 0) ? half_value(half_bits(rest)) : 0.0;
          arow[2 * D + t] = c;
          rest -= c;
        }
        for (int ks = 0; ks < KS; ks++)
          for (uint32_t kk = 0; kk < 32; kk++) {
            const uint32_t k = 32 * ks + kk;
            const double v = arow[k];
            const uint32_t lane = r + 16 * (kk >> 3), e = kk & 7;
            ap[(b * blk_bytes) / 2 + (size_t)ks * 512 + (size_t)lane * 8 + e] = half_bits(v);
          }
      }
    }
  }
  });
  if (Cs > 1) {  // the candidate test of every chunk uses the whole state's largest |a| and |konst|
    for (uint32_t st = 0; st < S; st++)
      for (int f = 0; f < 2; f++) {
        float mxv = 0.0f;
        for (uint32_t ch = 0; ch < Cs; ch++) {
          const float v = anorm[2 * (size_t)(st * Cs + ch) + f];
          if (!(v <= mxv)) mxv = v;
        }
        for (uint32_t ch = 0; ch < Cs; ch++) anorm[2 * (size_t)(st * Cs + ch) + f] = mxv;
      }
  }
  HIP_TRY(m->pf_apack.upload(reinterpret_cast<const unsigned char*>(ap.data()), ap.size() * 2));
  HIP_TRY(m->pf_anorm.upload(anorm.data(), anorm.size()));
  m->pf_groups = n_groups;
  m->pf_ks32 = KS;
  m->pf_ny = 0;
  return SR_OK;
}

int set_prefilter_splits(sr_model* m, uint32_t nx) {
  const uint32_t target_wgs = 16 * 768;
  uint32_t ny = (target_wgs + nx - 1) / std::max(1u, nx);
  ny = std::max(1u, std::min(ny, std::max(1u, m->pf_groups / 8)));
  if (ny >= 8) ny &= ~7u;
  // one table per split count, kept: launches of different sizes (the pieces of a corpus that is still being fed) alternate
  // between them without touching a table a queued kernel may still read
  auto it = m->pf_split_tabs.find(ny);
  if (it == m->pf_split_tabs.end() || !it->second || !it->second->p) {
    std::vector<uint32_t> sb(ny + 1);
    for (uint32_t y = 0; y <= ny; y++) sb[y] = (uint32_t)((uint64_t)m->pf_groups * y / ny);
    std::unique_ptr<DevBuf<uint32_t>> fresh(new DevBuf<uint32_t>());
    HIP_TRY(fresh->upload(sb.data(), sb.size()));  // (a failed upload leaves no entry behind: the next call tries again)
    it = m->pf_split_tabs.insert_or_assign(ny, std::move(fresh)).first;
  }
  m->pf_split_cur = it->second->p;
  m->pf_ny = ny;
  return SR_OK;
}

// choose the state-range split count for a launch over `nx` frame tiles and upload group-aligned ranges
int set_splits(sr_model* m, uint32_t nx) {
  // Equal-sized workgroups run in rounds of (2 per CU x 256 CUs); aim for ~32 rounds so the last,
  // partly filled round costs ~3 %, but keep >= 32 model blocks (~70 us of MFMA work) per workgroup
  // so the feature-tile prologue stays negligible.
  const uint32_t target_wgs = 32 * 512;
  uint32_t ny = (target_wgs + nx - 1) / std::max(1u, nx);
  ny = std::max(1u, std::min(ny, std::min(m->n_groups, std::max(1u, m->n_blocks / 32))));
  if (ny >= 8) ny &= ~7u;  // multiples of 8 enable the XCD-aware tile map
  auto it = m->split_tabs.find(ny);  // (kept per split count, see set_prefilter_splits)
  if (it == m->split_tabs.end() || !it->second || !it->second->p) {
    std::vector<uint32_t> sb(ny + 1);
    for (uint32_t y = 0; y <= ny; y++) {
      // balance by blocks, cut at group boundaries
      const uint64_t want = (uint64_t)m->n_blocks * y / ny;
      auto it = std::lower_bound(m->group_first_block.begin(), m->group_first_block.end(), (uint32_t)want);
      sb[y] = *it;
    }
    sb[0] = 0;
    sb[ny] = m->n_blocks;
    std::unique_ptr<DevBuf<uint32_t>> fresh(new DevBuf<uint32_t>());
    HIP_TRY(fresh->upload(sb.data(), sb.size()));
    it = m->split_tabs.insert_or_assign(ny, std::move(fresh)).first;
  }
  m->split_cur = it->second->p;
  m->split_ny = ny;
  return SR_OK;
}

int prof_begin(sr_model* m, hipStream_t s, int kind, EventPair* ep) {
  if (!m->profiling) return SR_OK;
  ep->kind = kind;
  HIP_TRY(hipEventCreate(&ep->a));
  HIP_TRY(hipEventCreate(&ep->b));
  HIP_TRY(hipEventRecord(ep->a, s));
  return SR_OK;
}
int prof_end(sr_model* m, hipStream_t s, EventPair* ep) {
  if (!m->profiling) return SR_OK;
  HIP_TRY(hipEventRecord(ep->b, s));
  m->events.push_back(*ep);
  return SR_OK;
}

// kernel-specific model packing (first use) and workspaces for scoring launches of up to n_max frames: growing a
// workspace frees the old one, which must not happen between launches that are still queued
// SR_GMM_DEFAULT for a dense table: the fastest kernel that gives the reference's results -- the bit-exact prefilter path for
// max-approx models; for sum scoring (Mixtures.cpp:719-728) the FP64-MFMA kernel with its fused -log sum exp epilogue, within
// SR_GMM_MFMA's 1e-9 (the direct form stays 1e-12 from the reference's libm in sum mode, at 3x the time: SR_GMM_EXACT on request).
int resolve_dense_kernel(const sr_model* m, int gmm_kernel) {
  if (gmm_kernel == SR_GMM_DEFAULT) gmm_kernel = m->max_approx ? SR_GMM_PREFILTER : SR_GMM_MFMA;
  if (gmm_kernel == SR_GMM_MFMA && m->ksteps == 0) gmm_kernel = SR_GMM_EXACT;  // dimension 64 .. 160: the exact kernel (tighter, not looser)
  return gmm_kernel;
}

static size_t defer_budget_bytes() {
  const char* e = getenv("SRGPU_DEFER_MB");
  return e ? (size_t)strtoull(e, nullptr, 10) << 20 : (size_t)32 << 30;
}

int reserve_scoring(sr_model* m, int gmm_kernel, uint64_t n_max) {
  gmm_kernel = resolve_dense_kernel(m, gmm_kernel);
  if ((gmm_kernel == SR_GMM_MFMA && !m->mfma_packed) || (gmm_kernel == SR_GMM_PREFILTER && !m->pf_packed)) {
    int rc = srhost::ensure_host_tables(m);
    if (rc) return rc;
  }
  if (gmm_kernel == SR_GMM_MFMA && !m->mfma_packed) {
    int rc = pack_model(m, m->h_dens_off.data(), m->h_means.data(), m->h_inv_vars.data(), m->h_norm.data(), m->h_logw.data());
    if (rc) return rc;
    m->mfma_packed = true;
  }
  if (gmm_kernel == SR_GMM_PREFILTER && !m->pf_packed) {
    int rc = pack_prefilter(m, m->h_dens_off.data(), m->h_means.data(), m->h_inv_vars.data(), m->h_norm.data(), m->h_logw.data());
    if (rc) return rc;
    m->pf_packed = true;
  }
  if (gmm_kernel == SR_GMM_PREFILTER && m->pf_ks32 > 0 && n_max > 0) {
    const uint64_t ldT = (n_max + 63) & ~(uint64_t)63;
    const size_t want_T = (size_t)ldT * m->pf_dp, want_mask = (size_t)((m->pf_groups + 1u) & ~1u) * n_max * 4;
    const size_t want_P = m->pf_dp != m->dim ? (size_t)n_max * m->pf_dp + 4 : 0;  // (+4: the rows are read in 16-byte pieces)
    GmmRefineArgs ra{};
    ra.n_frames = n_max; ra.dim = m->pf_dp; ra.n_pstates = m->pf_pstates; ra.n_slots = m->pf_slots;
    const size_t want_ring = gmm_refine_ring_words(ra);
    if (want_T > m->featsT.n || want_P > m->featsP.n || want_mask > m->pf_mask.n || want_ring > m->pf_ring.n) HIP_TRY(hipStreamSynchronize(m->s_gmm));
    HIP_TRY(m->featsT.ensure(want_T));
    if (want_P) HIP_TRY(m->featsP.ensure(want_P));
    {  // deferred leftovers: within SRGPU_DEFER_MB (default 32 GiB; 0 switches the route off)
      ra.chunks = m->pf_chunks;
      uint32_t cap; size_t n_e, n_c;
      gmm_refine_defer_layout(ra, defer_budget_bytes(), &cap, &n_e, &n_c);
      if (n_e > m->pf_defer.n || n_c > m->pf_defer_cnt.n) HIP_TRY(hipStreamSynchronize(m->s_gmm));
      if (n_e) { HIP_TRY(m->pf_defer.ensure(n_e)); HIP_TRY(m->pf_defer_cnt.ensure(n_c)); }
    }
    HIP_TRY(m->pf_mask.ensure(want_mask));  // the refinement reads group pairs
    HIP_TRY(m->pf_ring.ensure(want_ring));
  }
  return SR_OK;
}

// score frames [f_begin, f_end) of `feats` into `out` (device, row stride m->ld) on stream s_gmm
int launch_scoring(sr_model* m, const float* d_feats, uint64_t n_frames, int gmm_kernel, double* d_out) {
  if (n_frames == 0) return SR_OK;
  gmm_kernel = resolve_dense_kernel(m, gmm_kernel);
  EventPair ep{};
  {
    int rc = reserve_scoring(m, gmm_kernel, n_frames);
    if (rc) return rc;
  }
  if (gmm_kernel == SR_GMM_MFMA) {
    const uint32_t tile = gmm_mfma_frames_per_tile(m->ksteps);
    const uint32_t nx = (uint32_t)((n_frames + tile - 1) / tile);
    int rc = set_splits(m, nx);
    if (rc) return rc;
    GmmMfmaArgs a{};
    a.feats = d_feats; a.n_frames = n_frames; a.dim = m->dim;
    a.apack = m->apack.p; a.blk_meta = m->blk_meta.p; a.grp_state = m->grp_state.p; a.split_begin = m->split_cur;
    a.out = d_out; a.ld = m->ld; a.nx = nx; a.ny = m->split_ny;
    if ((rc = prof_begin(m, m->s_gmm, 0, &ep))) return rc;
    HIP_TRY(launch_gmm_mfma(a, m->ksteps, !m->max_approx, m->s_gmm));
    if ((rc = prof_end(m, m->s_gmm, &ep))) return rc;
  } else if (gmm_kernel == SR_GMM_PREFILTER && m->pf_ks32 > 0) {
    const uint32_t tile = gmm_prefilter_frames_per_tile();
    const uint32_t nx = (uint32_t)((n_frames + tile - 1) / tile);
    int rc = set_prefilter_splits(m, nx);
    if (rc) return rc;
    const uint64_t ldT = (n_frames + 63) & ~(uint64_t)63;
    GmmPrefilterArgs pa{};
    pa.feats = d_feats; pa.n_frames = n_frames; pa.dim = m->dim;
    pa.apack = m->pf_apack.p; pa.grp_anorm = m->pf_anorm.p; pa.split_begin = m->pf_split_cur;
    pa.mask = m->pf_mask.p; pa.nx = nx; pa.ny = m->pf_ny; pa.chunks = std::min(4u, m->pf_chunks);
    GmmRefineArgs ra{};
    const bool padded = m->pf_dp != m->dim;
    ra.feats = padded ? m->featsP.p : d_feats; ra.featsT = m->featsT.p; ra.n_frames = n_frames; ra.n_frames_ld = ldT; ra.dim = m->pf_dp;
    ra.n_pstates = m->pf_pstates; ra.chunks = m->pf_chunks;
    ra.n_dens_ps = m->pf_ndens.p; ra.rows = m->pf_rows.p; ra.n_slots = m->pf_slots;
    ra.mask = m->pf_mask.p; ra.out = d_out; ra.ld = m->ld;
    ra.n_refined = m->profiling ? m->pf_counter.p : nullptr;
    ra.ring = m->pf_ring.p;
    {
      uint32_t cap; size_t n_e, n_c;
      gmm_refine_defer_layout(ra, defer_budget_bytes(), &cap, &n_e, &n_c);
      if (cap && n_e <= m->pf_defer.n && n_c <= m->pf_defer_cnt.n) { ra.defer = m->pf_defer.p; ra.defer_cnt = m->pf_defer_cnt.p; ra.defer_cap = cap; }
    }
    if (m->profiling) m->prof.refined_pairs += n_frames * (uint64_t)m->n_states;
    EventPair ep_p{}, ep_r{};
    if ((rc = prof_begin(m, m->s_gmm, 0, &ep))) return rc;
    if ((rc = prof_begin(m, m->s_gmm, 2, &ep_p))) return rc;
    HIP_TRY(launch_transpose_feats(d_feats, n_frames, m->dim, m->pf_dp, ldT, m->featsT.p, padded ? m->featsP.p : nullptr, m->s_gmm));
    HIP_TRY(launch_gmm_prefilter(pa, m->pf_ks32, m->s_gmm));
    if ((rc = prof_end(m, m->s_gmm, &ep_p))) return rc;
    if ((rc = prof_begin(m, m->s_gmm, 3, &ep_r))) return rc;
    HIP_TRY(launch_gmm_refine(ra, m->s_gmm));
    if ((rc = prof_end(m, m->s_gmm, &ep_r))) return rc;
    if ((rc = prof_end(m, m->s_gmm, &ep))) return rc;
  } else if (gmm_kernel == SR_GMM_EXACT || gmm_kernel == SR_GMM_PREFILTER) {
    // (a model the prefilter cannot take -- sum scoring, > 256 densities per mixture (> 128 beyond dim 39), dim > 62, or a device that fails
    // the fp16 probes -- is scored by the exact kernel: same bits, FP64 VALU speed)
    GmmExactArgs a{};
    a.feats = d_feats; a.n_frames = n_frames; a.dim = m->dim; a.n_states = m->n_states;
    a.dens_off = m->dens_off.p; a.means = m->means.p; a.inv_vars = m->inv_vars.p; a.norm = m->norm.p; a.logw = m->logw.p;
    a.out = d_out; a.ld = m->ld;
    const uint32_t nx = (uint32_t)((n_frames + 255) / 256);
    uint32_t ny = std::max(1u, std::min(m->n_states, (4096 + nx - 1) / nx));
    a.states_per_split = (m->n_states + ny - 1) / ny;
    ny = (m->n_states + a.states_per_split - 1) / a.states_per_split;
    int rc;
    if ((rc = prof_begin(m, m->s_gmm, 0, &ep))) return rc;
    HIP_TRY(launch_gmm_exact(a, !m->max_approx, ny, m->s_gmm));
    if ((rc = prof_end(m, m->s_gmm, &ep))) return rc;
  } else {
    return fail(SR_EINVAL, "unknown gmm_kernel %d", gmm_kernel);
  }
  if (m->profiling) m->prof.gmm_flops += 4.0 * m->dim * (double)m->n_dens * (double)n_frames;
  return SR_OK;
}

// utterance chunks whose score table fits the workspace
struct Chunk { uint32_t u0, u1; uint64_t f0, f1; };
std::vector<Chunk> make_chunks(const sr_corpus* c, size_t chunk_frames) {
  std::vector<Chunk> out;
  // chunks of whole utterances, at most chunk_frames each (a longer utterance is a chunk of its own), and of about equal size
  // when several are needed: the search of chunk i runs beside the scoring of chunk i+1, and a short last chunk leaves the
  // long one's search without company (configs[4]: 147 ms per step with 16 + 3.4 GB chunks, 138.5 with 2 x 9.7)
  const uint64_t total = c->n_frames;
  const uint64_t n_target = std::max<uint64_t>(1, (total + chunk_frames - 1) / std::max<size_t>(1, chunk_frames));
  const uint64_t even = (total + n_target - 1) / n_target;
  uint32_t u = 0;
  while (u < c->n_utts) {
    uint32_t v = u + 1;
    while (v < c->n_utts && c->frame_off[v + 1] - c->frame_off[u] <= chunk_frames && c->frame_off[v] - c->frame_off[u] < even) v++;
    out.push_back({u, v, c->frame_off[u], c->frame_off[v]});
    u = v;
  }
  return out;
}

// One workgroup searches / aligns one utterance, and a launch has only a few workgroups per CU (configs[2]: 1000 utterances of
// 200..400 frames on 256 CUs): handed out in corpus order, the last CU finishes ~19 % after the average one.  Every launch range
// is therefore walked longest utterance first (ties in corpus order); results do not depend on the order.
int ensure_utt_order(sr_model* m, sr_corpus* c, const std::vector<Chunk>& chunks) {
  if (c->order_chunk_frames == m->chunk_frames && c->utt_order.p) return SR_OK;
  std::vector<uint32_t> order(c->n_utts);
  for (uint32_t u = 0; u < c->n_utts; u++) order[u] = u;
  for (const Chunk& ch : chunks)
    std::stable_sort(order.begin() + ch.u0, order.begin() + ch.u1, [&](uint32_t a, uint32_t b) {
      return c->frame_off[a + 1] - c->frame_off[a] > c->frame_off[b + 1] - c->frame_off[b];
    });
  HIP_TRY(c->utt_order.upload(order.data(), order.size()));
  c->order_chunk_frames = m->chunk_frames;
  return SR_OK;
}

// Scores frames [f0, f1) of the corpus into `table` (row 0 = frame f0).  While the feeder is still copying
// (sr_corpus_upload_async) the range is scored in four launches -- the first 1/24, up to the sixth, up to the half, the rest -- each
// waiting on the device only for its own pieces: scoring starts ~1 ms into the transfer and the rest of it hides behind
// the kernels (the feeder delivers 5-10 GB/s, scoring consumes 2 GB/s of features).  The search that follows sees one table.
int score_chunk(sr_model* m, sr_corpus* c, uint64_t f0, uint64_t f1, int gmm_kernel, double* table) {
  const uint64_t n = f1 - f0;
  uint64_t cuts[4] = {f1, f1, f1, f1};
  int n_cuts = 1;
  // (the first launch is what the transfer cannot hide behind: 1/24 of the chunk = 2 MB of configs[2]'s features, 0.3 ms of
  // staging copy + PCIe instead of the 1.1 ms the first sixth took)
  if (srhost::corpus_upload_in_flight(c) && n >= 24 * 2048) { cuts[0] = f0 + n / 24; cuts[1] = f0 + n / 6; cuts[2] = f0 + n / 2; n_cuts = 4; }
  int rc = reserve_scoring(m, gmm_kernel, n_cuts == 4 ? n - n / 2 : n);
  if (rc) return rc;
  uint64_t a = f0;
  for (int i = 0; i < n_cuts; i++) {
    if ((rc = srhost::corpus_ready(c, a, cuts[i], m->s_gmm))) return rc;
    if ((rc = launch_scoring(m, c->feats.p + a * m->dim, cuts[i] - a, gmm_kernel, table + (a - f0) * m->ld))) return rc;
    a = cuts[i];
  }
  return SR_OK;
}

int ensure_score_ws(sr_model* m, const std::vector<Chunk>& chunks) {
  uint64_t mx = 0;
  for (const Chunk& ch : chunks) mx = std::max(mx, ch.f1 - ch.f0);
  const int nbuf = chunks.size() > 1 ? 2 : 1;
  for (int i = 0; i < nbuf; i++) HIP_TRY(m->scores[i].ensure((size_t)mx * m->ld));
  return SR_OK;
}

}  // namespace
namespace srhost {
static void swap_spare(sr_corpus* c, CorpusSpare& sp) {
  c->feats.swap(sp.feats); c->d_frame_off.swap(sp.d_frame_off); c->utt_order.swap(sp.utt_order); c->out_words.swap(sp.out_words);
  c->out_count.swap(sp.out_count); c->out_flags.swap(sp.out_flags); c->tb_score.swap(sp.tb_score); c->tb_word.swap(sp.tb_word);
  c->tb_bkp.swap(sp.tb_bkp);
}
void corpus_register(sr_corpus* c) {
  sr_model* m = c->model;
  std::lock_guard<std::mutex> lk(m->spare_mu);
  m->corpora.push_back(c);
}
void corpus_adopt_spare(sr_corpus* c) {
  sr_model* m = c->model;
  std::unique_ptr<CorpusSpare> sp;
  { std::lock_guard<std::mutex> lk(m->spare_mu); sp = std::move(m->spare); }
  if (sp) swap_spare(c, *sp);  // (whatever the new corpus held -- nothing -- goes away with sp)
  c->order_chunk_frames = 0;   // the launch order is rebuilt for the new utterances
}
size_t corpus_spare_cap_bytes() {
  static const size_t cap = [] {
    const char* e = getenv("SRGPU_SPARE_MB");
    const long mb = e ? atol(e) : 256;
    return (size_t)(mb < 0 ? 0 : mb) << 20;
  }();
  return cap;
}
void corpus_donate_spare(sr_corpus* c) {
  sr_model* m = c->model;  // (srgpu.h: a corpus is destroyed BEFORE its model)
  if (!m) return;
  std::unique_ptr<CorpusSpare> sp(new CorpusSpare());
  swap_spare(c, *sp);
  std::lock_guard<std::mutex> lk(m->spare_mu);
  m->corpora.erase(std::remove(m->corpora.begin(), m->corpora.end(), c), m->corpora.end());
  if (sp->bytes() > corpus_spare_cap_bytes()) return;  // too big to keep: freed on return
  if (!m->spare) m->spare = std::move(sp);  // (else: one spare set is kept, this one is freed on return)
}
}  // namespace srhost
namespace {

int check_model(const sr_model* m) {
  if (!m) return fail(SR_EINVAL, "null model handle");
  HIP_TRY(hipSetDevice(m->device));
  return SR_OK;
}

}  // namespace

extern "C" {

const char* sr_last_error(void) { return g_err; }

int sr_abi_version(void) { return SR_ABI_VERSION; }

int sr_device_count(int* count) {
  return guarded(__func__, [&]() -> int {
  if (!count) return fail(SR_EINVAL, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(SR_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return SR_OK;
  });
}

}  // extern "C"

namespace srhost {

// A model handle with everything but the four parameter tables: device checks, streams, events, dens_off, chunk sizes.
// sr_model_create uploads host tables into it; the device-side finalize (em_finalize.hip) fills them where they are.
int model_shell(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off, int max_approx, sr_model** out) {
  *out = nullptr;
  if (!dens_off) return fail(SR_EINVAL, "null model table");
  if (dim == 0 || n_states == 0) return fail(SR_EINVAL, "dim and n_states must be positive");
  // (0 beyond dimension 63: no dense FP64-MFMA instantiation -- SR_GMM_MFMA is then answered by the exact kernel, resolve_dense_kernel)
  const int ks = gmm_mfma_ksteps_for_dim(dim);
  if (dim > 160) return fail(SR_ELIMIT, "dim %u unsupported (max 160: the exact kernel keeps a workgroup's frames in LDS)", dim);
  if (dens_off[0] != 0) return fail(SR_EINVAL, "dens_off[0] must be 0");
  for (uint32_t s = 0; s < n_states; s++)
    if (dens_off[s + 1] < dens_off[s]) return fail(SR_EINVAL, "dens_off must be non-decreasing (state %u)", s);
  const uint64_t C = dens_off[n_states];
  if (C >= (1ull << 31)) return fail(SR_ELIMIT, "too many densities");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SR_ENODEV, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(SR_EINVAL, "device %d out of range (0..%d)", device, ndev - 1);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SR_ENODEV, "device %d is %s; libsrgpu is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  HIP_TRY(hipSetDevice(device));
  sr_model* m = new sr_model();
  std::unique_ptr<sr_model, int (*)(sr_model*)> own(m, sr_model_destroy);  // released on success; an exception frees it
  m->device = device; m->dim = dim; m->n_states = n_states; m->n_dens = C; m->max_approx = max_approx != 0;
  m->ksteps = ks;
  m->ld = (n_states + 7u) & ~7u;  // 64-byte rows pieces for the kernels that write 8 states per thread
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&m->s_gmm, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&m->s_search, hipStreamNonBlocking)) != hipSuccess)
    return fail(SR_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
  for (int i = 0; i < 2; i++)
    if ((e = hipEventCreateWithFlags(&m->ev_scored[i], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&m->ev_consumed[i], hipEventDisableTiming)) != hipSuccess)
      return fail(SR_EHIP, "hipEventCreate: %s", hipGetErrorString(e));
  if ((e = m->dens_off.upload(dens_off, n_states + 1)) != hipSuccess) return fail(SR_EHIP, "model upload: %s", hipGetErrorString(e));
  uint32_t mx = 0;
  for (uint32_t s = 0; s < n_states; s++) mx = std::max(mx, dens_off[s + 1] - dens_off[s]);
  m->max_dens = std::max(1u, mx);
  m->h_dens_off.assign(dens_off, dens_off + n_states + 1);
  const char* env = getenv("SRGPU_SCORE_CHUNK_MB");
  // score workspace per chunk: 48 GiB by default, at most a sixth of the device's memory (two such buffers only when a
  // corpus needs more than one chunk, and the candidate masks of a chunk take up to as much again).  Bigger chunks mean
  // fewer, longer launches (configs[4]'s 19.4 GB table: 136.8 ms per step in one chunk, 147.1 in 16 + 3.4 GB) and MI355X has
  // 288 GB of HBM to spend.
  size_t chunk_bytes = (size_t)49152 << 20;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) chunk_bytes = std::min(chunk_bytes, total_b / 6);
  }
  if (env) chunk_bytes = (size_t)atol(env) << 20;
  m->chunk_frames = std::max<size_t>(1, chunk_bytes / ((size_t)m->ld * sizeof(double)));
  // the refinement kernel addresses the transposed feature copy of a chunk with 32-bit buffer offsets
  m->chunk_frames = std::min<size_t>(m->chunk_frames, ((size_t)1 << 32) / (4 * (size_t)dim) - 128);
  if (const char* ov = getenv("SRGPU_OVERLAP")) m->overlap = atoi(ov) != 0;
  *out = own.release();
  return SR_OK;
}

// host copies of the parameter tables, for the kernel-specific packings: a model finalised on the device has none until
// a packing is asked for
int ensure_host_tables(sr_model* m) {
  const size_t C = m->n_dens, D = m->dim;
  if (m->h_norm.size() == C && m->h_means.size() == C * D) return SR_OK;
  m->h_means.resize(C * D); m->h_inv_vars.resize(C * D); m->h_norm.resize(C); m->h_logw.resize(C);
  if (C == 0) return SR_OK;
  HIP_TRY(hipMemcpy(m->h_means.data(), m->means.p, C * D * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(m->h_inv_vars.data(), m->inv_vars.p, C * D * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(m->h_norm.data(), m->norm.p, C * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(m->h_logw.data(), m->logw.p, C * sizeof(double), hipMemcpyDeviceToHost));
  return SR_OK;
}

// Can an emission cost of this model be negative (or not a number)?  The search kernels' collapsed word-boundary transition rests on
// costs >= 0 (Recognizer.cpp:143,173: the early-out is inert then); a model that can break it is searched by the variant that replays
// the early-out (viterbi_words.hip, NEG) instead of being flagged utterance by utterance.  A density's score is
// fl(fl(norm + dist / 2) - logw) with dist >= 0 whenever every inverse variance is >= 0, and rounding is monotone, so
// fl(norm - logw) >= 0 for every density rules negative costs out (a variance <= 0 or NaN makes norm -inf or NaN: caught).  Sum
// scoring (Mixtures.cpp:719-728) lies up to log(#densities) below the smallest density score.
int may_go_negative(sr_model* m, bool* out) {
  if (m->neg_possible < 0) {
    const size_t C = m->n_dens;
    std::vector<double> norm_l, logw_l;
    const double *norm = m->h_norm.data(), *logw = m->h_logw.data();
    if (m->h_norm.size() != C || m->h_logw.size() != C) {
      norm_l.resize(C); logw_l.resize(C);
      if (C) {
        HIP_TRY(hipMemcpy(norm_l.data(), m->norm.p, C * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(logw_l.data(), m->logw.p, C * sizeof(double), hipMemcpyDeviceToHost));
      }
      norm = norm_l.data(); logw = logw_l.data();
    }
    bool neg = false;
    for (uint32_t s = 0; s < m->n_states && !neg; s++) {
      const uint32_t c0 = m->h_dens_off[s], c1 = m->h_dens_off[s + 1];
      const double slack = m->max_approx ? 0.0 : std::log((double)std::max(1u, c1 - c0)) + 1e-6;
      for (uint32_t c = c0; c < c1; c++)
        if (!(norm[c] - logw[c] - slack >= 0.0)) { neg = true; break; }
    }
    // tables handed in through sr_model_create are taken as they are: a negative or NaN inverse variance there
    if (!neg && m->h_inv_vars.size() == C * (size_t)m->dim)
      for (double iv : m->h_inv_vars)
        if (!(iv >= 0.0)) { neg = true; break; }
    m->neg_possible = neg ? 1 : 0;
  }
  *out = m->neg_possible == 1;
  return SR_OK;
}

}  // namespace srhost

extern "C" {

int sr_model_create(int device, uint32_t dim, uint32_t n_states, const uint32_t* dens_off, const double* means,
                    const double* inv_vars, const double* norm, const double* logw, int max_approx, sr_model** out) {
  return guarded(__func__, [&]() -> int {
  if (!out) return fail(SR_EINVAL, "out is null");
  *out = nullptr;
  if (!dens_off || !means || !inv_vars || !norm || !logw) return fail(SR_EINVAL, "null model table");
  sr_model* m = nullptr;
  int rc = srhost::model_shell(device, dim, n_states, dens_off, max_approx, &m);
  if (rc != SR_OK) return rc;
  std::unique_ptr<sr_model, int (*)(sr_model*)> own(m, sr_model_destroy);
  const uint64_t C = m->n_dens;
  hipError_t e;
  if ((e = m->means.upload(means, C * dim)) != hipSuccess || (e = m->inv_vars.upload(inv_vars, C * dim)) != hipSuccess ||
      (e = m->norm.upload(norm, C)) != hipSuccess || (e = m->logw.upload(logw, C)) != hipSuccess)
    return fail(SR_EHIP, "model upload: %s", hipGetErrorString(e));
  m->h_means.assign(means, means + C * dim);
  m->h_inv_vars.assign(inv_vars, inv_vars + C * dim);
  m->h_norm.assign(norm, norm + C);
  m->h_logw.assign(logw, logw + C);
  std::vector<uint32_t> ident(C);
  std::iota(ident.begin(), ident.end(), 0u);
  m->n_mean = m->n_var = (uint32_t)C;
  m->h_dens_mean = ident; m->h_dens_var = ident;
  if ((e = m->dens_mean.upload(ident.data(), C)) != hipSuccess || (e = m->dens_var.upload(ident.data(), C)) != hipSuccess)
    return fail(SR_EHIP, "tying upload: %s", hipGetErrorString(e));
  *out = own.release();
  return SR_OK;
  });
}

int sr_model_destroy(sr_model* m) {
  return guarded(__func__, [&]() -> int {
  if (!m) return SR_OK;
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  for (auto& ep : m->events) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  {  // corpora that outlive the model (srgpu.h asks for the other order) must not reach into it when they are destroyed
    std::lock_guard<std::mutex> lk(m->spare_mu);
    for (sr_corpus* c : m->corpora) c->model = nullptr;
    m->corpora.clear();
  }
  m->dens_mean.release(); m->dens_var.release();
  m->dens_off.release(); m->means.release(); m->inv_vars.release(); m->norm.release(); m->logw.release();
  m->apack.release(); m->blk_meta.release(); m->grp_state.release();
  m->scores[0].release(); m->scores[1].release();
  for (int i = 0; i < 2; i++) {
    if (m->ev_scored[i]) (void)hipEventDestroy(m->ev_scored[i]);
    if (m->ev_consumed[i]) (void)hipEventDestroy(m->ev_consumed[i]);
  }
  for (int b = 0; b < 2; b++) {
    if (m->staging_free[b]) (void)hipEventDestroy(m->staging_free[b]);
    if (m->staging[b]) (void)hipHostFree(m->staging[b]);
  }
  if (m->s_copy) (void)hipStreamDestroy(m->s_copy);
  if (m->s_gmm) (void)hipStreamDestroy(m->s_gmm);
  if (m->s_search) (void)hipStreamDestroy(m->s_search);
  delete m;
  return SR_OK;
  });
}

int sr_model_info(const sr_model* m, uint32_t* dim, uint32_t* n_states, uint64_t* n_densities) {
  return guarded(__func__, [&]() -> int {
  if (!m) return fail(SR_EINVAL, "null model handle");
  if (dim) *dim = m->dim;
  if (n_states) *n_states = m->n_states;
  if (n_densities) *n_densities = m->n_dens;
  return SR_OK;
  });
}

int sr_corpus_upload(sr_model* m, const float* feats, const uint64_t* frame_off, uint32_t n_utts, sr_corpus** out) {
  return guarded(__func__, [&]() -> int {
  if (!out) return fail(SR_EINVAL, "out is null");
  *out = nullptr;
  int rc = check_model(m);
  if (rc) return rc;
  if (!frame_off) return fail(SR_EINVAL, "frame_off is null");
  if (frame_off[0] != 0) return fail(SR_EINVAL, "frame_off[0] must be 0");
  for (uint32_t u = 0; u < n_utts; u++) {
    if (frame_off[u + 1] < frame_off[u]) return fail(SR_EINVAL, "frame_off must be non-decreasing (utterance %u)", u);
    if (frame_off[u + 1] - frame_off[u] > 65535)
      return fail(SR_ELIMIT, "utterance %u has %llu frames; back pointers are 16 bit like the reference's Book::bkp (max 65535)",
                  u, (unsigned long long)(frame_off[u + 1] - frame_off[u]));
  }
  const uint64_t F = frame_off[n_utts];
  if (F > 0 && !feats) return fail(SR_EINVAL, "feats is null");
  sr_corpus* c = new sr_corpus();
  std::unique_ptr<sr_corpus, int (*)(sr_corpus*)> own(c, sr_corpus_destroy);
  c->model = m; c->n_utts = n_utts; c->n_frames = F;
  srhost::corpus_register(c);
  c->frame_off.assign(frame_off, frame_off + n_utts + 1);
  srhost::corpus_adopt_spare(c);
  hipError_t e;
  if ((e = c->feats.ensure((size_t)F * m->dim + 64)) != hipSuccess ||
      (F > 0 && (e = hipMemcpy(c->feats.p, feats, (size_t)F * m->dim * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) ||
      (e = c->d_frame_off.upload(frame_off, n_utts + 1)) != hipSuccess)
    return fail(SR_EHIP, "corpus upload: %s", hipGetErrorString(e));
  *out = own.release();
  return SR_OK;
  });
}

int sr_model_trim(sr_model* m) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  std::unique_ptr<CorpusSpare> sp;
  { std::lock_guard<std::mutex> lk(m->spare_mu); sp = std::move(m->spare); }
  // ... and the refinement's deferred-leftover segments (23 GB at configs[4]); the next scoring call allocates them again
  m->pf_defer.release(); m->pf_defer_cnt.release();
  return SR_OK;  // (sp's buffers are freed here)
  });
}

int sr_corpus_destroy(sr_corpus* c) {
  return guarded(__func__, [&]() -> int {
  if (!c) return SR_OK;
  srhost::feeder_join(c);  // an upload still in flight borrows the caller's buffer and writes into c->feats
  if (c->model) { (void)hipSetDevice(c->model->device); (void)hipDeviceSynchronize(); }
  srhost::corpus_donate_spare(c);  // the search-path buffers stay with the model for its next corpus
  c->feats.release(); c->d_frame_off.release(); c->tb_score.release(); c->tb_word.release(); c->tb_bkp.release();
  c->out_words.release(); c->out_count.release(); c->out_flags.release(); c->automata.release(); c->out_states.release();
  c->aut_off.release(); c->bp_off.release(); c->al_blk_frame0.release(); c->al_list_off.release(); c->al_states.release();
  c->al_blk_frames.release(); c->al_blk_list.release(); c->backptr.release(); c->out_cost.release(); c->path_scores.release();
  c->pair_off.release(); c->pair_frame.release(); c->key_mean.release(); c->key_var.release(); c->iota.release();
  c->keys_sorted.release(); c->pairs_sorted.release(); c->row_begin.release(); c->pair_w.release(); c->acc_mean.release(); c->acc_var.release();
  c->w_mean.release(); c->w_var.release(); c->sort_temp.release();
  delete c;
  return SR_OK;
  });
}

int sr_score_corpus(sr_model* m, sr_corpus* c, int gmm_kernel, double* out) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  if (!c || c->model != m) return fail(SR_EINVAL, "corpus does not belong to this model");
  if (!out && c->n_frames) return fail(SR_EINVAL, "out is null");
  // chunk over plain frame ranges; utterance boundaries do not matter for scoring
  const uint64_t F = c->n_frames;
  const size_t step = m->chunk_frames;
  HIP_TRY(m->scores[0].ensure((size_t)std::min<uint64_t>(F, step) * m->ld));
  for (uint64_t f = 0; f < F; f += step) {
    const uint64_t n = std::min<uint64_t>(step, F - f);
    if ((rc = srhost::corpus_ready(c, f, f + n, m->s_gmm))) return rc;
    if ((rc = launch_scoring(m, c->feats.p + f * m->dim, n, gmm_kernel, m->scores[0].p))) return rc;
    HIP_TRY(hipStreamSynchronize(m->s_gmm));
    HIP_TRY(hipMemcpy2D(out + f * m->n_states, (size_t)m->n_states * sizeof(double), m->scores[0].p,
                        (size_t)m->ld * sizeof(double), (size_t)m->n_states * sizeof(double), (size_t)n,
                        hipMemcpyDeviceToHost));
  }
  if (m->profiling) m->prof.frames += F;
  return SR_OK;
  });
}

int sr_score_frames(sr_model* m, const float* feats, uint64_t n_frames, int gmm_kernel, double* out) {
  return guarded(__func__, [&]() -> int {
  const uint64_t off[2] = {0, n_frames};
  int rc = check_model(m);
  if (rc) return rc;
  if (n_frames > 0 && (!feats || !out)) return fail(SR_EINVAL, "%s is null", feats ? "out" : "feats");
  if (n_frames > (std::numeric_limits<size_t>::max() / sizeof(float) - 64) / m->dim)
    return fail(SR_ELIMIT, "n_frames %llu x dim %u overflows the address space", (unsigned long long)n_frames, m->dim);
  // scoring has no 16-bit frame limit: bypass the per-utterance check by uploading directly
  sr_corpus* c = new sr_corpus();
  std::unique_ptr<sr_corpus, int (*)(sr_corpus*)> own(c, sr_corpus_destroy);
  c->model = m; c->n_utts = 1; c->n_frames = n_frames; c->frame_off.assign(off, off + 2);
  srhost::corpus_register(c);
  hipError_t e;
  if ((e = c->feats.ensure((size_t)n_frames * m->dim + 64)) != hipSuccess ||
      (n_frames > 0 && (e = hipMemcpy(c->feats.p, feats, (size_t)n_frames * m->dim * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess))
    return fail(SR_EHIP, "feature upload: %s", hipGetErrorString(e));
  return sr_score_corpus(m, c, gmm_kernel, out);
  });
}

int sr_lexicon_create(sr_model* m, uint32_t n_words, const uint32_t* word_off, const uint16_t* automaton,
                      uint32_t silence_idx, const double tdp[3], uint16_t silence_state, sr_lexicon** out) {
  return guarded(__func__, [&]() -> int {
  if (!out) return fail(SR_EINVAL, "out is null");
  *out = nullptr;
  int rc = check_model(m);
  if (rc) return rc;
  if (!word_off || !automaton || !tdp) return fail(SR_EINVAL, "null lexicon table");
  if (n_words == 0 || n_words > 65535) return fail(SR_ELIMIT, "n_words must be 1..65535 (Book::word is 16 bit)");
  if (silence_idx >= n_words) return fail(SR_EINVAL, "silence_idx out of range");
  if (word_off[0] != 0) return fail(SR_EINVAL, "word_off[0] must be 0");
  uint32_t max_pos = 0;
  for (uint32_t w = 0; w < n_words; w++) {
    if (word_off[w + 1] <= word_off[w]) return fail(SR_EINVAL, "word %u has no positions", w);
    max_pos = std::max(max_pos, word_off[w + 1] - word_off[w]);
  }
  // the reference addresses hypotheses as word*max_pos + pos and lets entry into position 1 alias the
  // next word's position 0 when every word has a single position (Recognizer.cpp:139); not reproduced
  if (max_pos < 2) return fail(SR_ELIMIT, "lexicon needs at least one word with two or more positions");
  const uint32_t P = word_off[n_words];
  if (P > decode_big_max_slots()) return fail(SR_ELIMIT, "%u trellis positions exceed the decoder's limit of %u", P, decode_big_max_slots());
  std::vector<uint32_t> info(P), sword(P), wend(n_words);
  for (uint32_t w = 0; w < n_words; w++) {
    const uint32_t b = word_off[w], n = word_off[w + 1] - b;
    const uint32_t first = automaton[b];
    for (uint32_t k = 0; k < n; k++) {
      const uint32_t st = automaton[b + k];
      if (st >= m->n_states) return fail(SR_EINVAL, "word %u position %u: state %u >= n_states %u", w, k, st, m->n_states);
      uint32_t f = st;
      if (k == 0) f |= 1u << 16;
      if (k == 1) f |= 1u << 17;
      if (k == n - 1) f |= 1u << 18;
      if (st == silence_state) f |= 1u << 19;
      if (w == silence_idx) f |= 1u << 20;
      if (first == silence_state) f |= 1u << 21;
      if (n == 1) f |= 1u << 22;
      info[b + k] = f;
      sword[b + k] = w;
    }
    wend[w] = b + n - 1;
  }
  // ---- type-sorted network for the fast kernel (viterbi_fast.hip) ---------------------------------------------
  // key = kind | silence-state << 3 | silence-word << 4 | first-state-is-silence << 5; kinds: 0 entry position 0,
  // 1 single-position word, 2 entry position 1, 3 position 1 that is also the word end, 4 middle, 5 word end
  std::vector<uint32_t> key(P);
  for (uint32_t p = 0; p < P; p++) {
    const uint32_t f = info[p];
    const bool pos0 = f & (1u << 16), pos1 = f & (1u << 17), end = f & (1u << 18);
    const uint32_t kind = pos0 ? (end ? 1u : 0u) : pos1 ? (end ? 3u : 2u) : (end ? 5u : 4u);
    key[p] = kind | ((f >> 19 & 1u) << 3) | ((f >> 20 & 1u) << 4) | ((f >> 21 & 1u) << 5);
  }
  std::vector<uint32_t> new_id(P), f_state, f_pred, f_orig, f_type, f_word;
  for (uint32_t k = 0; k < 64; k++) {
    bool any = false;
    for (uint32_t p = 0; p < P; p++) {
      if (key[p] != k) continue;
      any = true;
      new_id[p] = (uint32_t)f_orig.size();
      f_orig.push_back(p);  // completed below
    }
    if (!any) continue;
    while (f_orig.size() % 64) f_orig.push_back(0xFFFFFFFFu);
    f_type.resize(f_orig.size() / 64, k);
  }
  uint32_t Pn = (uint32_t)f_orig.size();
  // beyond what the LDS holds (type padding included) the search runs from a global workspace (decode_big_kernel): no fast net
  const bool big = Pn > decode_max_slots();
  if (big) { Pn = 0; f_orig.clear(); f_type.clear(); }
  f_state.assign(Pn, 0); f_pred.assign(Pn, 0); f_word.assign(Pn, 0);
  for (uint32_t q = 0; q < Pn; q++) {
    const uint32_t p = f_orig[q];
    if (p == 0xFFFFFFFFu) { f_pred[q] = q | (q << 16); continue; }
    const uint32_t w = sword[p], base = word_off[w], k = p - base;
    f_state[q] = info[p] & 0xFFFFu;
    f_word[q] = w;
    f_pred[q] = (k >= 1 ? new_id[p - 1] : q) | ((k >= 2 ? new_id[p - 2] : q) << 16);
    f_orig[q] = p | (base << 16);
  }
  // ---- word by word for the word-per-lane kernel (viterbi_words.hip): every word at most four positions --------------------
  std::vector<uint32_t> w_info, w_order;
  std::vector<uint2> w_states;
  uint32_t plain_len = 0, w_nw = 0, w_nt = 0, w_general = 0;
  if (max_pos <= 4 && n_words <= decode_words_max_words()) {
    w_info.resize(n_words); w_states.resize(n_words);
    uint32_t hist[5] = {0, 0, 0, 0, 0};
    for (uint32_t w = 0; w < n_words; w++) {
      const uint32_t b = word_off[w], n = word_off[w + 1] - b;
      uint32_t f = n, st[4] = {0, 0, 0, 0};
      if (w == silence_idx) f |= 8u;
      if (automaton[b] == silence_state) f |= 16u;
      for (uint32_t k = 0; k < n; k++) {
        st[k] = automaton[b + k];
        if (st[k] == silence_state) f |= 1u << (8 + k);
      }
      w_info[w] = f;
      w_states[w] = make_uint2(st[0] | (st[1] << 16), st[2] | (st[3] << 16));
      if (f == n) hist[n]++;  // no flags
    }
    // plain words: the commonest flag-free length of 2..4 positions.  Every kind fills whole groups of 64 lane slots; slot
    // s = tid + k * nt belongs to group s / 64 = k * n_waves + wave.  Plain and single groups are dealt to the first waves, nw
    // per wave; a general group -- three times the instructions of a plain one -- gets a wave of its own (its other groups
    // stay empty), because a frame lasts as long as its heaviest wave (viterbi_words.hip).
    plain_len = 3;
    for (uint32_t n = 2; n <= 4; n++) if (hist[n] > hist[plain_len]) plain_len = n;
    std::vector<uint32_t> kinds[3];  // plain, single, general
    for (uint32_t w = 0; w < n_words; w++)
      kinds[w_info[w] == plain_len ? 0 : (w_info[w] & 7u) == 1u ? 1 : 2].push_back(w);
    const uint32_t g_plain = ((uint32_t)kinds[0].size() + 63) / 64, g_single = ((uint32_t)kinds[1].size() + 63) / 64,
                   g_gen = ((uint32_t)kinds[2].size() + 63) / 64;
    w_general = g_gen ? 1u : 0u;
    // words per lane: the fewest that leave a workgroup of at most 8 waves -- two of them share a CU then (128 registers per
    // lane each), and while one waits at its barrier the other computes --, else the fewest that fit 16 waves
    auto waves_for = [&](uint32_t nw) { return (g_plain + g_single + nw - 1) / nw + g_gen; };
    uint32_t waves = 0;
    for (w_nw = 1; w_nw <= 3 && waves_for(w_nw) > 8; w_nw++) {}
    if (w_nw > 3) for (w_nw = 1; w_nw <= 3 && waves_for(w_nw) > 16; w_nw++) {}
    if (w_nw <= 3) {
      waves = waves_for(w_nw);
      w_nt = waves * 64;
      w_order.assign((size_t)w_nw * w_nt, 0xFFFFFFFFu);
      auto put_group = [&](uint32_t wave, uint32_t k, const std::vector<uint32_t>& words, uint32_t g) {
        for (uint32_t i = 0; i < 64 && (size_t)g * 64 + i < words.size(); i++) w_order[(size_t)k * w_nt + wave * 64 + i] = words[(size_t)g * 64 + i];
      };
      uint32_t seq = 0;  // plain groups, then single groups: wave seq / nw, slot row seq % nw
      for (uint32_t g = 0; g < g_plain; g++, seq++) put_group(seq / w_nw, seq % w_nw, kinds[0], g);
      for (uint32_t g = 0; g < g_single; g++, seq++) put_group(seq / w_nw, seq % w_nw, kinds[1], g);
      for (uint32_t g = 0; g < g_gen; g++) put_group(waves - g_gen + g, 0, kinds[2], g);
    } else {
      plain_len = 0;  // more groups than a workgroup has room for: the slot-per-lane kernel
    }
  }
  sr_lexicon* l = new sr_lexicon();
  std::unique_ptr<sr_lexicon, int (*)(sr_lexicon*)> own(l, sr_lexicon_destroy);
  l->w_plain_len = plain_len; l->w_nw = w_nw; l->w_nt = w_nt; l->w_general = w_general;
  l->f_n = Pn; l->f_init = new_id[0]; l->f_init_end = (info[0] >> 18) & 1u; l->big = big;
  l->model = m; l->n_words = n_words; l->n_slots = P; l->silence_idx = silence_idx; l->silence_state = silence_state;
  l->tdp[0] = tdp[0]; l->tdp[1] = tdp[1]; l->tdp[2] = tdp[2];
  hipError_t e;
  if ((e = l->slot_info.upload(info.data(), P)) != hipSuccess || (e = l->slot_word.upload(sword.data(), P)) != hipSuccess ||
      (e = l->word_end_slot.upload(wend.data(), n_words)) != hipSuccess || (e = l->f_state.upload(f_state.data(), Pn)) != hipSuccess ||
      (e = l->f_pred.upload(f_pred.data(), Pn)) != hipSuccess || (e = l->f_orig.upload(f_orig.data(), Pn)) != hipSuccess ||
      (e = l->f_type.upload(f_type.data(), Pn / 64)) != hipSuccess || (e = l->f_word.upload(f_word.data(), Pn)) != hipSuccess ||
      (e = l->w_info.upload(w_info.data(), w_info.size())) != hipSuccess || (e = l->w_states.upload(w_states.data(), w_states.size())) != hipSuccess ||
      (e = l->w_order.upload(w_order.data(), w_order.size())) != hipSuccess)
    return fail(SR_EHIP, "lexicon upload: %s", hipGetErrorString(e));
  *out = own.release();
  return SR_OK;
  });
}

int sr_lexicon_describe(const sr_lexicon* l, char* out, size_t cap) {
  return guarded(__func__, [&]() -> int {
  if (!l || !out || cap == 0) return fail(SR_EINVAL, "null argument");
  DecodeArgs da{};
  da.words.info = l->w_plain_len ? l->w_info.p : nullptr;
  da.ld = l->model ? l->model->ld : 0;
  if (l->big && decode_words_applies(da)) snprintf(out, cap, "words %u x %u plain %u%s (replay: big, %u positions)", l->w_nw, l->w_nt, l->w_plain_len, l->w_general ? " general" : "", l->n_slots);
  else if (l->big) snprintf(out, cap, "big (%u positions: hypotheses in device memory)", l->n_slots);
  else if (decode_words_applies(da)) snprintf(out, cap, "words %u x %u plain %u%s", l->w_nw, l->w_nt, l->w_plain_len, l->w_general ? " general" : "");
  else snprintf(out, cap, "slots (%u type-padded positions)", l->f_n);
  return SR_OK;
  });
}

int sr_lexicon_destroy(sr_lexicon* l) {
  return guarded(__func__, [&]() -> int {
  if (!l) return SR_OK;
  if (l->model) { (void)hipSetDevice(l->model->device); (void)hipDeviceSynchronize(); }
  l->slot_info.release(); l->slot_word.release(); l->word_end_slot.release();
  l->f_state.release(); l->f_pred.release(); l->f_orig.release(); l->f_type.release(); l->f_word.release(); l->w_info.release(); l->w_states.release(); l->w_order.release();
  delete l;
  return SR_OK;
  });
}

// Words of utterance u start at out_words[frame_off[u]] on the device, out_count[u] of them; a kernel that could not walk an
// utterance's traceback (traceback.h) has raised kFlagCorrupt instead of reporting words: SR_ECORRUPT, naming the first one.
static int gather_words(sr_corpus* c, uint32_t* out_words, uint64_t* out_word_off) {
  const uint32_t U = c->n_utts;
  const uint64_t F = c->n_frames;
  std::vector<uint32_t> counts(U), flags(U), dev_words(F);
  if (U) HIP_TRY(hipMemcpy(counts.data(), c->out_count.p, sizeof(uint32_t) * U, hipMemcpyDeviceToHost));
  if (U) HIP_TRY(hipMemcpy(flags.data(), c->out_flags.p, sizeof(uint32_t) * U, hipMemcpyDeviceToHost));
  if (F) HIP_TRY(hipMemcpy(dev_words.data(), c->out_words.p, sizeof(uint32_t) * F, hipMemcpyDeviceToHost));
  uint64_t w = 0;
  out_word_off[0] = 0;
  for (uint32_t u = 0; u < U; u++) {
    if (flags[u] & kFlagCorrupt) return fail(SR_ECORRUPT, "utterance %u: the traceback does not walk back to frame 0 (Recognizer.cpp:222-231)", u);
    const uint64_t b = c->frame_off[u], T = c->frame_off[u + 1] - b;
    if (counts[u] > T) return fail(SR_ECORRUPT, "utterance %u: %u words for %llu frames", u, counts[u], (unsigned long long)T);
    for (uint32_t i = 0; i < counts[u]; i++) out_words[w++] = dev_words[b + i];
    out_word_off[u + 1] = w;
  }
  return SR_OK;
}

int sr_recognize_corpus(sr_model* m, sr_corpus* c, sr_lexicon* l, const sr_search_params* p, uint32_t* out_words,
                        uint64_t* out_word_off, double* tb_score, uint16_t* tb_word, uint16_t* tb_bkp) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  if (!c || c->model != m) return fail(SR_EINVAL, "corpus does not belong to this model");
  if (!l || l->model != m) return fail(SR_EINVAL, "lexicon does not belong to this model");
  if (!p || !out_word_off || (!out_words && c->n_frames)) return fail(SR_EINVAL, "null argument");
  const uint32_t U = c->n_utts;
  const uint64_t F = c->n_frames;
  HIP_TRY(c->tb_score.ensure(F + U));
  HIP_TRY(c->tb_word.ensure(F + U));
  HIP_TRY(c->tb_bkp.ensure(F + U));
  HIP_TRY(c->out_words.ensure(F));
  HIP_TRY(c->out_count.ensure(U));
  HIP_TRY(c->out_flags.ensure(U));
  HIP_TRY(hipMemsetAsync(c->out_flags.p, 0, sizeof(uint32_t) * std::max(1u, U), m->s_gmm));
  const std::vector<Chunk> chunks = make_chunks(c, m->chunk_frames);
  if ((rc = ensure_score_ws(m, chunks))) return rc;
  if ((rc = ensure_utt_order(m, c, chunks))) return rc;

  if (l->big) {  // hypothesis arrays of the utterances in flight
    uint32_t most = 0;
    for (const Chunk& ch : chunks) most = std::max(most, ch.u1 - ch.u0);
    HIP_TRY(c->big_ws.ensure((size_t)most * decode_big_workspace(l->n_slots)));
  }
  DecodeArgs da{};
  da.net.n_slots = l->n_slots; da.net.n_words = l->n_words;
  da.net.slot_info = l->slot_info.p; da.net.slot_word = l->slot_word.p; da.net.word_end_slot = l->word_end_slot.p;
  da.net.silence_word = l->silence_idx; da.net.silence_state = l->silence_state;
  da.net.tdp_loop = l->tdp[0]; da.net.tdp_forward = l->tdp[1]; da.net.tdp_skip = l->tdp[2];
  da.fast.n_slots = l->f_n; da.fast.state = l->f_state.p; da.fast.pred = l->f_pred.p; da.fast.orig = l->f_orig.p;
  da.fast.chunk_type = l->f_type.p; da.fast.word = l->f_word.p; da.fast.init_slot = l->f_init; da.fast.init_is_end = l->f_init_end;
  da.words.info = l->w_plain_len ? l->w_info.p : nullptr; da.words.states = l->w_states.p; da.words.order = l->w_order.p;
  da.words.plain_len = l->w_plain_len; da.words.nw = l->w_nw; da.words.nt = l->w_nt; da.words.has_general = l->w_general;
  da.words.init_is_end = l->f_init_end;
  da.ld = m->ld; da.frame_off = c->d_frame_off.p; da.utt_order = c->utt_order.p;
  da.am_threshold = p->am_threshold; da.word_penalty = p->word_penalty;
  if (p->flags & ~(SR_SEARCH_GENERAL_KERNEL | SR_SEARCH_SLOT_KERNEL)) return fail(SR_EINVAL, "unknown sr_search_params.flags 0x%x", (unsigned)p->flags);
  da.force_general = (p->flags & SR_SEARCH_GENERAL_KERNEL) ? 1u : 0u;
  da.force_slots = (p->flags & SR_SEARCH_SLOT_KERNEL) ? 1u : 0u;
  {
    bool neg = false;
    if ((rc = srhost::may_go_negative(m, &neg))) return rc;
    da.exact_negative = neg ? 1u : 0u;
  }
  da.tb_score = c->tb_score.p; da.tb_word = c->tb_word.p; da.tb_bkp = c->tb_bkp.p;
  da.out_words = c->out_words.p; da.out_count = c->out_count.p; da.out_flags = c->out_flags.p;

  // chunk i is scored on s_gmm into buffer i&1 while chunk i-1 is searched on s_search
  hipStream_t s_search = m->overlap ? m->s_search : m->s_gmm;
  for (size_t i = 0; i < chunks.size(); i++) {
    const Chunk& ch = chunks[i];
    const int buf = (int)(i & 1);
    if (i >= 2) HIP_TRY(hipStreamWaitEvent(m->s_gmm, m->ev_consumed[buf], 0));
    if ((rc = score_chunk(m, c, ch.f0, ch.f1, p->gmm_kernel, m->scores[buf].p))) return rc;
    HIP_TRY(hipEventRecord(m->ev_scored[buf], m->s_gmm));
    HIP_TRY(hipStreamWaitEvent(s_search, m->ev_scored[buf], 0));
    da.scores = m->scores[buf].p; da.frame_base = ch.f0; da.utt_first = ch.u0; da.n_utts = ch.u1 - ch.u0;
    EventPair ep{};
    if ((rc = prof_begin(m, s_search, 1, &ep))) return rc;
    if (l->big && !da.force_general && !da.force_slots && decode_words_applies(da)) {
      // more than 8192 type-padded positions, but words of at most four: the word-per-lane kernel does not depend on the slot count
      // (ADVICE r3); what it flags (a negative emission cost) is redone by the device-memory kernel, which exits at once elsewhere
      HIP_TRY(launch_decode_words(da, s_search));
      da.only_flagged = 1;
      HIP_TRY(launch_decode_big(da, c->big_ws.p, s_search));
      da.only_flagged = 0;
    } else {
      HIP_TRY(l->big ? launch_decode_big(da, c->big_ws.p, s_search) : launch_decode(da, s_search));
    }
    if ((rc = prof_end(m, s_search, &ep))) return rc;
    HIP_TRY(hipEventRecord(m->ev_consumed[buf], s_search));
    if (m->profiling) m->prof.search_bytes += (8.0 * m->n_states + 4.0 * l->n_slots) * (double)(ch.f1 - ch.f0);
  }
  HIP_TRY(hipStreamSynchronize(m->s_search));
  HIP_TRY(hipStreamSynchronize(m->s_gmm));

  if ((rc = gather_words(c, out_words, out_word_off))) return rc;
  if (tb_score) HIP_TRY(hipMemcpy(tb_score, c->tb_score.p, sizeof(double) * (F + U), hipMemcpyDeviceToHost));
  if (tb_word) HIP_TRY(hipMemcpy(tb_word, c->tb_word.p, sizeof(uint16_t) * (F + U), hipMemcpyDeviceToHost));
  if (tb_bkp) HIP_TRY(hipMemcpy(tb_bkp, c->tb_bkp.p, sizeof(uint16_t) * (F + U), hipMemcpyDeviceToHost));
  if (m->profiling) m->prof.frames += F;
  return SR_OK;
  });
}

int sr_traceback_corpus(sr_model* m, sr_corpus* c, sr_lexicon* l, const uint16_t* tb_word, const uint16_t* tb_bkp,
                        uint32_t* out_words, uint64_t* out_word_off) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  if (!c || c->model != m) return fail(SR_EINVAL, "corpus does not belong to this model");
  if (!l || l->model != m) return fail(SR_EINVAL, "lexicon does not belong to this model");
  if (!tb_word || !tb_bkp || !out_word_off || (!out_words && c->n_frames)) return fail(SR_EINVAL, "null argument");
  const uint32_t U = c->n_utts;
  const uint64_t F = c->n_frames;
  HIP_TRY(c->tb_word.ensure(F + U));
  HIP_TRY(c->tb_bkp.ensure(F + U));
  HIP_TRY(c->out_words.ensure(F));
  HIP_TRY(c->out_count.ensure(U));
  HIP_TRY(c->out_flags.ensure(U));
  HIP_TRY(hipMemcpyAsync(c->tb_word.p, tb_word, sizeof(uint16_t) * (F + U), hipMemcpyHostToDevice, m->s_gmm));
  HIP_TRY(hipMemcpyAsync(c->tb_bkp.p, tb_bkp, sizeof(uint16_t) * (F + U), hipMemcpyHostToDevice, m->s_gmm));
  HIP_TRY(launch_traceback(c->d_frame_off.p, U, c->tb_word.p, c->tb_bkp.p, l->silence_idx, l->n_words, c->out_words.p,
                           c->out_count.p, c->out_flags.p, m->s_gmm));
  HIP_TRY(hipStreamSynchronize(m->s_gmm));
  return gather_words(c, out_words, out_word_off);
  });
}

int sr_traceback_words(uint32_t n_frames, const uint16_t* tb_word, const uint16_t* tb_bkp, uint32_t silence_word, uint32_t n_words,
                       uint32_t* out_words, uint32_t* out_count) {
  return guarded(__func__, [&]() -> int {
  if (!tb_word || !tb_bkp || !out_count || (!out_words && n_frames)) return fail(SR_EINVAL, "null argument");
  *out_count = 0;
  const uint32_t n = walk_traceback(
      n_frames, silence_word, n_words, [&](uint32_t t) -> uint32_t { return tb_word[t]; },
      [&](uint32_t t) -> uint32_t { return tb_bkp[t]; }, out_words, n_frames);
  if (n == kTbCorrupt) return fail(SR_ECORRUPT, "not a traceback: a back pointer that does not fall, or a word outside the lexicon");
  *out_count = n;
  return SR_OK;
  });
}

int sr_bigram_create(sr_model* m, uint32_t n_words, const uint32_t* word_off, const uint16_t* mixtures,
                     uint32_t silence_word, const float* lm, const float tdp[8], sr_bigram** out) {
  return guarded(__func__, [&]() -> int {
  if (!out) return fail(SR_EINVAL, "out is null");
  *out = nullptr;
  int rc = check_model(m);
  if (rc) return rc;
  if (!word_off || !mixtures || !lm || !tdp) return fail(SR_EINVAL, "null argument");
  const uint32_t W = n_words;
  if (W == 0 || silence_word >= W) return fail(SR_EINVAL, "silence word %u out of range (%u words)", silence_word, W);
  if (W > bigram_max_words()) return fail(SR_ELIMIT, "%u words: the bigram search handles at most %u", W, bigram_max_words());
  if (word_off[0] != 0) return fail(SR_EINVAL, "word_off[0] must be 0");
  for (uint32_t w = 0; w < W; w++) {
    if (word_off[w + 1] <= word_off[w]) return fail(SR_EINVAL, "word %u has no states", w);
    for (uint32_t i = word_off[w]; i < word_off[w + 1]; i++)
      if (mixtures[i] >= m->n_states) return fail(SR_EINVAL, "word %u: mixture %u out of range", w, mixtures[i]);
  }
  // slots: words 0..W-1, then the silence copy of every word (Teaching::LinearSearch: silenceCopy :202-205)
  const uint32_t n_sil = word_off[silence_word + 1] - word_off[silence_word];
  std::vector<uint32_t> slot_off(2 * (size_t)W + 1, 0), slot_mix(2 * (size_t)W);
  for (uint32_t a = 0; a < 2 * W; a++) {
    const uint32_t aw = a < W ? a : silence_word;
    slot_off[a + 1] = slot_off[a] + (a < W ? word_off[a + 1] - word_off[a] : n_sil);
    slot_mix[a] = word_off[aw];
  }
  const uint32_t P2 = slot_off[2 * W];
  // per dense position: emission state | flags << 16 and the slot (the state update runs one thread per position)
  std::vector<uint32_t> pos_info(P2);
  std::vector<uint32_t> pos_slot(P2);
  for (uint32_t a = 0; a < 2 * W; a++) {
    const uint32_t n = slot_off[a + 1] - slot_off[a], sil = (a >= W || a == silence_word) ? 8u : 0u;
    for (uint32_t k = 0; k < n; k++) {
      const uint32_t flags = (k == 0 ? 1u : 0u) | (k == 1 ? 2u : 0u) | (k == n - 1 ? 4u : 0u) | sil;
      pos_info[slot_off[a] + k] = (uint32_t)mixtures[slot_mix[a] + k] | (flags << 16);
      pos_slot[slot_off[a] + k] = a;
    }
  }
  {  // the dense LDS image, or -- short words -- the register layout, which needs less (viterbi_bigram.hip)
    BigramArgs probe{};
    probe.n_words = W; probe.ld = m->ld; probe.silence_states = n_sil;
    for (uint32_t a2 = 0; a2 < 2 * W; a2++) probe.max_slot_states = std::max(probe.max_slot_states, slot_off[a2 + 1] - slot_off[a2]);
    if (bigram_lds_bytes(W, P2) > 160 * 1024 && !bigram_register_layout(probe))
      return fail(SR_ELIMIT, "lexicon too large for the bigram search's LDS image (%zu bytes > 160 KiB)", bigram_lds_bytes(W, P2));
  }
  std::vector<float> lmT((size_t)W * W);
  for (uint32_t w = 0; w < W; w++)
    for (uint32_t h = 0; h < W; h++) lmT[(size_t)h * W + w] = lm[(size_t)w * W + h];
  std::vector<float> rowmin(W, std::numeric_limits<float>::infinity()), rowmax(W, -std::numeric_limits<float>::infinity());
  for (uint32_t h = 0; h < W; h++)
    for (uint32_t w = 0; w < W; w++) {
      if (w == silence_word) continue;  // no transition into silence through the LM (LinearSearch.cc:231)
      const float v = lmT[(size_t)h * W + w];
      // NaN entries: the bounds become NaN and the skip test fails safe (nothing is skipped against a NaN bound)
      rowmin[h] = (v < rowmin[h] || v != v) ? v : rowmin[h];
      rowmax[h] = (v > rowmax[h] || v != v) ? v : rowmax[h];
    }
  sr_bigram* b = new sr_bigram();
  std::unique_ptr<sr_bigram, int (*)(sr_bigram*)> own(b, sr_bigram_destroy);
  b->model = m; b->n_words = W; b->silence = silence_word; b->n_positions = P2;
  for (uint32_t a2 = 0; a2 < 2 * W; a2++) b->max_slot_states = std::max(b->max_slot_states, slot_off[a2 + 1] - slot_off[a2]);
  b->silence_states = n_sil;
  for (uint32_t w = 0; w < W; w++)  // (words only: a silence copy has the silence word's one state in the register layout)
    if (slot_off[w + 1] - slot_off[w] >= 4) b->row4_mask |= 1u << (w / 1024u);
  memcpy(b->tdp, tdp, sizeof(b->tdp));
  hipError_t e;
  if ((e = b->slot_off.upload(slot_off.data(), slot_off.size())) != hipSuccess ||
      (e = b->slot_mix.upload(slot_mix.data(), slot_mix.size())) != hipSuccess ||
      (e = b->mixtures.upload(mixtures, word_off[W])) != hipSuccess || (e = b->lmT.upload(lmT.data(), lmT.size())) != hipSuccess ||
      (e = b->lm_rowmin.upload(rowmin.data(), W)) != hipSuccess || (e = b->lm_rowmax.upload(rowmax.data(), W)) != hipSuccess ||
      (e = b->pos_info.upload(pos_info.data(), P2)) != hipSuccess || (e = b->pos_slot.upload(pos_slot.data(), P2)) != hipSuccess)
    return fail(SR_EHIP, "bigram upload: %s", hipGetErrorString(e));
  *out = own.release();
  return SR_OK;
  });
}

int sr_bigram_destroy(sr_bigram* b) {
  return guarded(__func__, [&]() -> int {
  if (!b) return SR_OK;
  if (b->model) { (void)hipSetDevice(b->model->device); (void)hipDeviceSynchronize(); }
  b->slot_off.release(); b->slot_mix.release(); b->mixtures.release(); b->pos_info.release(); b->pos_slot.release(); b->lmT.release(); b->lm_rowmin.release(); b->lm_rowmax.release();
  b->we_slot.release(); b->we_bp.release(); b->we_score.release(); b->book.release(); b->book_off.release();
  b->out_word.release(); b->out_time.release(); b->out_score.release(); b->out_count.release(); b->out_flags.release();
  delete b;
  return SR_OK;
  });
}

int sr_recognize_bigram_corpus(sr_model* m, sr_corpus* c, sr_bigram* b, const sr_bigram_params* p, uint32_t* out_word,
                               float* out_score, uint32_t* out_time, uint64_t* out_off) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  if (!c || c->model != m) return fail(SR_EINVAL, "corpus does not belong to this model");
  if (!b || b->model != m) return fail(SR_EINVAL, "bigram search net does not belong to this model");
  if (!p || !out_off || ((!out_word || !out_score || !out_time) && c->n_frames)) return fail(SR_EINVAL, "null argument");
  const uint32_t U = c->n_utts, W = b->n_words;
  const uint64_t F = c->n_frames;
  const std::vector<Chunk> chunks = make_chunks(c, m->chunk_frames);
  if ((rc = ensure_score_ws(m, chunks))) return rc;
  // traceback book: 2 start entries + (word ends kept per frame <= W) * T per utterance
  const uint64_t per_frame = p->max_word_ends ? std::min<uint64_t>(p->max_word_ends, W) : W;
  std::vector<uint64_t> book_off(U + 1, 0);
  uint32_t max_chunk_utts = 0;
  for (uint32_t u = 0; u < U; u++) book_off[u + 1] = book_off[u] + 2 + per_frame * (c->frame_off[u + 1] - c->frame_off[u]);
  for (const Chunk& ch : chunks) max_chunk_utts = std::max(max_chunk_utts, ch.u1 - ch.u0);
  HIP_TRY(b->book.ensure(book_off[U]));
  HIP_TRY(b->book_off.upload(book_off.data(), book_off.size()));
  HIP_TRY(b->we_slot.ensure((size_t)max_chunk_utts * 4 * W));
  HIP_TRY(b->we_bp.ensure((size_t)max_chunk_utts * 4 * W));
  HIP_TRY(b->we_score.ensure((size_t)max_chunk_utts * 4 * W));
  HIP_TRY(b->out_word.ensure(F + U));
  HIP_TRY(b->out_time.ensure(F + U));
  HIP_TRY(b->out_score.ensure(F + U));
  HIP_TRY(b->out_count.ensure(U));
  HIP_TRY(b->out_flags.ensure(U));

  BigramArgs ba{};
  if ((rc = ensure_utt_order(m, c, chunks))) return rc;
  ba.ld = m->ld; ba.frame_off = c->d_frame_off.p; ba.utt_order = c->utt_order.p;
  ba.n_words = W; ba.silence = b->silence; ba.n_positions = b->n_positions;
  ba.slot_off = b->slot_off.p; ba.slot_mix = b->slot_mix.p; ba.mixtures = b->mixtures.p; ba.pos_info = b->pos_info.p; ba.pos_slot = b->pos_slot.p; ba.lmT = b->lmT.p; ba.lm_rowmin = b->lm_rowmin.p; ba.lm_rowmax = b->lm_rowmax.p;
  memcpy(ba.tdp, b->tdp, sizeof(ba.tdp));
  ba.ac_pruning = p->acoustic_pruning; ba.lm_pruning = p->lm_pruning;
  if (p->flags & ~SR_BIGRAM_DENSE_STATES) return fail(SR_EINVAL, "unknown sr_bigram_params.flags 0x%x", (unsigned)p->flags);
  ba.max_slot_states = b->max_slot_states; ba.silence_states = b->silence_states; ba.row4_mask = b->row4_mask;
  ba.dense_states = (p->flags & SR_BIGRAM_DENSE_STATES) ? 1u : 0u;
  if (!bigram_register_layout(ba) && bigram_lds_bytes(W, b->n_positions) > 160 * 1024)
    return fail(SR_ELIMIT, "this lexicon runs in the register layout only (its dense LDS image would take %zu bytes > 160 KiB)", bigram_lds_bytes(W, b->n_positions));
  ba.we_slot = b->we_slot.p; ba.we_bp = b->we_bp.p; ba.we_score = b->we_score.p;
  ba.book = b->book.p; ba.book_off = b->book_off.p;
  ba.out_word = b->out_word.p; ba.out_score = b->out_score.p; ba.out_time = b->out_time.p;
  ba.out_count = b->out_count.p; ba.out_flags = b->out_flags.p;

  hipStream_t s_search = m->overlap ? m->s_search : m->s_gmm;
  for (size_t i = 0; i < chunks.size(); i++) {
    const Chunk& ch = chunks[i];
    const int buf = (int)(i & 1);
    if (i >= 2) HIP_TRY(hipStreamWaitEvent(m->s_gmm, m->ev_consumed[buf], 0));
    if ((rc = score_chunk(m, c, ch.f0, ch.f1, p->gmm_kernel, m->scores[buf].p))) return rc;
    HIP_TRY(hipEventRecord(m->ev_scored[buf], m->s_gmm));
    HIP_TRY(hipStreamWaitEvent(s_search, m->ev_scored[buf], 0));
    ba.scores = m->scores[buf].p; ba.frame_base = ch.f0; ba.utt_first = ch.u0; ba.n_utts = ch.u1 - ch.u0;
    EventPair ep{};
    if ((rc = prof_begin(m, s_search, 1, &ep))) return rc;
    HIP_TRY(launch_bigram(ba, s_search));
    if ((rc = prof_end(m, s_search, &ep))) return rc;
    HIP_TRY(hipEventRecord(m->ev_consumed[buf], s_search));
    // SURVEY 8(d)'s decoder model, 8 S + 4 P bytes per frame, with P = the bigram search's positions (words + their silence copies)
    if (m->profiling) m->prof.search_bytes += (8.0 * m->n_states + 4.0 * b->n_positions) * (double)(ch.f1 - ch.f0);
  }
  HIP_TRY(hipStreamSynchronize(m->s_search));
  HIP_TRY(hipStreamSynchronize(m->s_gmm));

  std::vector<uint32_t> counts(U), flags(U), dw(F + U), dt(F + U);
  std::vector<float> ds(F + U);
  if (U) {
    HIP_TRY(hipMemcpy(counts.data(), b->out_count.p, sizeof(uint32_t) * U, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(flags.data(), b->out_flags.p, sizeof(uint32_t) * U, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dw.data(), b->out_word.p, sizeof(uint32_t) * (F + U), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dt.data(), b->out_time.p, sizeof(uint32_t) * (F + U), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ds.data(), b->out_score.p, sizeof(float) * (F + U), hipMemcpyDeviceToHost));
  }
  uint64_t n = 0;
  out_off[0] = 0;
  for (uint32_t u = 0; u < U; u++) {
    if (flags[u]) return fail(SR_ELIMIT, "utterance %u: more than %llu word ends per frame (sr_bigram_params.max_word_ends)", u,
                              (unsigned long long)per_frame);
    const uint64_t o = c->frame_off[u] + u;
    for (uint32_t i = 0; i < counts[u]; i++, n++) { out_word[n] = dw[o + i]; out_score[n] = ds[o + i]; out_time[n] = dt[o + i]; }
    out_off[u + 1] = n;
  }
  if (m->profiling) m->prof.frames += F;
  return SR_OK;
  });
}

int sr_recognize_batch(sr_model* m, sr_lexicon* l, const sr_search_params* p, const float* feats,
                       const uint64_t* frame_off, uint32_t n_utts, uint32_t* out_words, uint64_t* out_word_off) {
  return guarded(__func__, [&]() -> int {
  // the feeder copies while the first chunks are being scored (feeder.cpp); the corpus goes away with `own`, which also
  // ends the borrowing of the caller's buffer
  sr_corpus* c = nullptr;
  int rc = sr_corpus_upload_async(m, feats, frame_off, n_utts, &c);
  if (rc) return rc;
  std::unique_ptr<sr_corpus, int (*)(sr_corpus*)> own(c, sr_corpus_destroy);
  return sr_recognize_corpus(m, c, l, p, out_words, out_word_off, nullptr, nullptr, nullptr);
  });
}

static int align_common(sr_model* m, sr_corpus* c, const uint16_t* automata, const uint64_t* aut_off, const double tdp[3],
                        uint16_t silence_state, double thr, bool pruned, int gmm_kernel, uint16_t* out_states,
                        double* out_cost) {
  int rc = check_model(m);
  if (rc) return rc;
  if (gmm_kernel == SR_GMM_DEFAULT) gmm_kernel = SR_GMM_PREFILTER;  // the aligner scores only its automaton's states: listed, bit-exact
  if (!c || c->model != m) return fail(SR_EINVAL, "corpus does not belong to this model");
  if (!automata || !aut_off || !tdp || !out_states || !out_cost) return fail(SR_EINVAL, "null argument");
  const uint32_t U = c->n_utts;
  const uint64_t F = c->n_frames;
  std::vector<uint64_t> bp_off(U + 1, 0);
  uint32_t max_n = 1;
  for (uint32_t u = 0; u < U; u++) {
    const uint64_t N = aut_off[u + 1] - aut_off[u], T = c->frame_off[u + 1] - c->frame_off[u];
    if (N < 1 || T < 1) return fail(SR_EINVAL, "utterance %u: automaton and utterance must be non-empty", u);
    if (!pruned && N > T)
      return fail(SR_EINVAL, "utterance %u: automaton length %llu exceeds %llu frames (Aligner::align_sequence_full indexes "
                  "its T-sized cost arrays by position)", u, (unsigned long long)N, (unsigned long long)T);
    if (N > align_max_positions()) return fail(SR_ELIMIT, "utterance %u: automaton length %llu exceeds %u", u, (unsigned long long)N, align_max_positions());
    for (uint64_t i = aut_off[u]; i < aut_off[u + 1]; i++)
      if (automata[i] >= m->n_states) return fail(SR_EINVAL, "utterance %u: automaton state %u >= n_states", u, automata[i]);
    max_n = std::max<uint32_t>(max_n, (uint32_t)N);
    bp_off[u + 1] = bp_off[u] + N * T;
  }
  HIP_TRY(c->automata.upload(automata, aut_off[U]));
  HIP_TRY(c->aut_off.upload(aut_off, U + 1));
  HIP_TRY(c->bp_off.upload(bp_off.data(), U + 1));
  HIP_TRY(c->backptr.ensure(bp_off[U]));
  HIP_TRY(c->out_states.ensure(F));
  HIP_TRY(c->out_cost.ensure(U));
  const std::vector<Chunk> chunks = make_chunks(c, m->chunk_frames);
  if ((rc = ensure_score_ws(m, chunks))) return rc;
  // The aligner reads scores of its automaton's states only (N of S): with a bit-exact kernel requested, score just
  // those (frame, state) pairs -- the direct-form kernel in listed mode, the same bits as the dense table would hold.
  // (SR_GMM_MFMA keeps the dense FP64-MFMA table: its rounding differs.)
  const bool listed = gmm_kernel != SR_GMM_MFMA;
  std::vector<uint32_t> blk_first_of_utt(U + 1, 0);
  if (listed) {
    const uint32_t fpb = (uint32_t)gmm_exact_frames_per_block();
    std::vector<uint32_t> list_off(U + 1, 0), list_states, blk_frames, blk_list;
    std::vector<uint64_t> blk_frame0;
    for (uint32_t u = 0; u < U; u++) {
      std::vector<uint32_t> st(automata + aut_off[u], automata + aut_off[u + 1]);
      std::sort(st.begin(), st.end());
      st.erase(std::unique(st.begin(), st.end()), st.end());
      list_states.insert(list_states.end(), st.begin(), st.end());
      list_off[u + 1] = (uint32_t)list_states.size();
      for (uint64_t f = c->frame_off[u]; f < c->frame_off[u + 1]; f += fpb) {
        blk_frame0.push_back(f);
        blk_frames.push_back((uint32_t)std::min<uint64_t>(fpb, c->frame_off[u + 1] - f));
        blk_list.push_back(u);
      }
      blk_first_of_utt[u + 1] = (uint32_t)blk_frame0.size();
    }
    HIP_TRY(c->al_list_off.upload(list_off.data(), list_off.size()));
    HIP_TRY(c->al_states.upload(list_states.data(), list_states.size()));
    HIP_TRY(c->al_blk_frame0.upload(blk_frame0.data(), blk_frame0.size()));
    HIP_TRY(c->al_blk_frames.upload(blk_frames.data(), blk_frames.size()));
    HIP_TRY(c->al_blk_list.upload(blk_list.data(), blk_list.size()));
  }
  AlignArgs aa{};
  if ((rc = ensure_utt_order(m, c, chunks))) return rc;
  aa.ld = m->ld; aa.frame_off = c->d_frame_off.p; aa.utt_order = c->utt_order.p; aa.automata = c->automata.p; aa.aut_off = c->aut_off.p;
  aa.tdp_loop = tdp[0]; aa.tdp_forward = tdp[1]; aa.tdp_skip = tdp[2]; aa.silence_state = silence_state;
  aa.pruning_threshold = thr; aa.backptr = c->backptr.p; aa.bp_off = c->bp_off.p; aa.max_positions = max_n;
  aa.out_states = c->out_states.p; aa.out_cost = c->out_cost.p;
  for (size_t i = 0; i < chunks.size(); i++) {
    const Chunk& ch = chunks[i];
    const int buf = (int)(i & 1);
    if (i >= 2) HIP_TRY(hipStreamWaitEvent(m->s_gmm, m->ev_consumed[buf], 0));
    if ((rc = srhost::corpus_ready(c, ch.f0, ch.f1, m->s_gmm))) return rc;
    if (listed) {
      GmmExactArgs ga{};
      ga.feats = c->feats.p; ga.n_frames = c->n_frames; ga.dim = m->dim; ga.n_states = m->n_states;
      ga.dens_off = m->dens_off.p; ga.means = m->means.p; ga.inv_vars = m->inv_vars.p; ga.norm = m->norm.p; ga.logw = m->logw.p;
      ga.out = m->scores[buf].p; ga.ld = m->ld;
      GmmExactList gl{};
      gl.blk_frame0 = c->al_blk_frame0.p; gl.blk_frames = c->al_blk_frames.p; gl.blk_list = c->al_blk_list.p;
      gl.list_off = c->al_list_off.p; gl.states = c->al_states.p;
      gl.blk_first = blk_first_of_utt[ch.u0]; gl.frame_base = ch.f0;
      EventPair eg{};
      if ((rc = prof_begin(m, m->s_gmm, 0, &eg))) return rc;
      HIP_TRY(launch_gmm_exact_listed(ga, !m->max_approx, gl, blk_first_of_utt[ch.u1] - blk_first_of_utt[ch.u0], m->s_gmm));
      if ((rc = prof_end(m, m->s_gmm, &eg))) return rc;
    } else if ((rc = launch_scoring(m, c->feats.p + ch.f0 * m->dim, ch.f1 - ch.f0, gmm_kernel, m->scores[buf].p))) {
      return rc;
    }
    HIP_TRY(hipEventRecord(m->ev_scored[buf], m->s_gmm));
    HIP_TRY(hipStreamWaitEvent(m->s_search, m->ev_scored[buf], 0));
    aa.scores = m->scores[buf].p; aa.frame_base = ch.f0; aa.utt_first = ch.u0; aa.n_utts = ch.u1 - ch.u0;
    EventPair ep{};
    if ((rc = prof_begin(m, m->s_search, 1, &ep))) return rc;
    HIP_TRY(pruned ? launch_align_pruned(aa, m->s_search) : launch_align_full(aa, m->s_search));
    if ((rc = prof_end(m, m->s_search, &ep))) return rc;
    HIP_TRY(hipEventRecord(m->ev_consumed[buf], m->s_search));
  }
  HIP_TRY(hipStreamSynchronize(m->s_search));
  HIP_TRY(hipStreamSynchronize(m->s_gmm));
  if (F) HIP_TRY(hipMemcpy(out_states, c->out_states.p, sizeof(uint16_t) * F, hipMemcpyDeviceToHost));
  if (U) HIP_TRY(hipMemcpy(out_cost, c->out_cost.p, sizeof(double) * U, hipMemcpyDeviceToHost));
  if (m->profiling) {
    m->prof.frames += F;
    for (uint32_t u = 0; u < U; u++) m->prof.search_bytes += 9.0 * (double)bp_off[u + 1] - 9.0 * (double)bp_off[u];
  }
  return SR_OK;
}

int sr_align_corpus(sr_model* m, sr_corpus* c, const uint16_t* automata, const uint64_t* aut_off, const double tdp[3],
                    uint16_t silence_state, int gmm_kernel, uint16_t* out_states, double* out_cost) {
  return guarded(__func__, [&]() -> int {
  return align_common(m, c, automata, aut_off, tdp, silence_state, 0.0, false, gmm_kernel, out_states, out_cost);
  });
}

int sr_align_corpus_pruned(sr_model* m, sr_corpus* c, const uint16_t* automata, const uint64_t* aut_off,
                           const double tdp[3], uint16_t silence_state, double pruning_threshold, int gmm_kernel,
                           uint16_t* out_states, double* out_cost) {
  return guarded(__func__, [&]() -> int {
  return align_common(m, c, automata, aut_off, tdp, silence_state, pruning_threshold, true, gmm_kernel, out_states, out_cost);
  });
}

int sr_path_scores_corpus(sr_model* m, sr_corpus* c, const uint16_t* states, int gmm_kernel, double* out) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  if (gmm_kernel == SR_GMM_DEFAULT) gmm_kernel = SR_GMM_PREFILTER;  // (frame, state) pairs scored directly, bit-exact
  if (!c || c->model != m) return fail(SR_EINVAL, "corpus does not belong to this model");
  const uint64_t F = c->n_frames;
  if (F == 0) return SR_OK;
  if (!states || !out) return fail(SR_EINVAL, "null argument");
  for (uint64_t f = 0; f < F; f++)
    if (states[f] >= m->n_states) return fail(SR_EINVAL, "frame %llu: state %u >= n_states", (unsigned long long)f, states[f]);
  if ((rc = srhost::corpus_ready(c, 0, F, m->s_gmm))) return rc;
  HIP_TRY(c->out_states.upload(states, F));
  HIP_TRY(c->path_scores.ensure(F));
  if (gmm_kernel != SR_GMM_MFMA) {  // bit-exact kernels: score the F (frame, state) pairs directly, no dense table
    EmArgs a{};
    a.feats = c->feats.p; a.n_frames = F; a.dim = m->dim; a.states = c->out_states.p;
    a.dens_off = m->dens_off.p; a.means = m->means.p; a.inv_vars = m->inv_vars.p; a.norm = m->norm.p; a.logw = m->logw.p;
    a.max_approx = m->max_approx;
    HIP_TRY(launch_path_scores_direct(a, c->path_scores.p, m->s_gmm));
    HIP_TRY(hipStreamSynchronize(m->s_gmm));
    HIP_TRY(hipMemcpy(out, c->path_scores.p, sizeof(double) * F, hipMemcpyDeviceToHost));
    if (m->profiling) m->prof.frames += F;
    return SR_OK;
  }
  const size_t step = m->chunk_frames;
  HIP_TRY(m->scores[0].ensure((size_t)std::min<uint64_t>(F, step) * m->ld));
  for (uint64_t f = 0; f < F; f += step) {
    const uint64_t n = std::min<uint64_t>(step, F - f);
    if ((rc = launch_scoring(m, c->feats.p + f * m->dim, n, gmm_kernel, m->scores[0].p))) return rc;
    HIP_TRY(launch_path_scores(m->scores[0].p, m->ld, f, f, f + n, c->out_states.p, c->path_scores.p, m->s_gmm));
  }
  HIP_TRY(hipStreamSynchronize(m->s_gmm));
  HIP_TRY(hipMemcpy(out, c->path_scores.p, sizeof(double) * F, hipMemcpyDeviceToHost));
  if (m->profiling) m->prof.frames += F;
  return SR_OK;
  });
}

int sr_model_set_tying(sr_model* m, uint32_t n_mean, uint32_t n_var, const uint32_t* dens_mean, const uint32_t* dens_var) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  if (!dens_mean || !dens_var) return fail(SR_EINVAL, "null tying table");
  for (uint64_t c = 0; c < m->n_dens; c++)
    if (dens_mean[c] >= n_mean || dens_var[c] >= n_var) return fail(SR_EINVAL, "density %llu: tying index out of range", (unsigned long long)c);
  HIP_TRY(m->dens_mean.upload(dens_mean, m->n_dens));
  HIP_TRY(m->dens_var.upload(dens_var, m->n_dens));
  m->n_mean = n_mean; m->n_var = n_var;
  m->h_dens_mean.assign(dens_mean, dens_mean + m->n_dens);
  m->h_dens_var.assign(dens_var, dens_var + m->n_dens);
  return SR_OK;
  });
}

int sr_model_topology(const sr_model* m, uint32_t* dens_off, uint32_t* dens_mean, uint32_t* dens_var) {
  return guarded(__func__, [&]() -> int {
  if (!m) return fail(SR_EINVAL, "null model handle");
  if (dens_off) std::copy(m->h_dens_off.begin(), m->h_dens_off.end(), dens_off);
  if (dens_mean) std::copy(m->h_dens_mean.begin(), m->h_dens_mean.end(), dens_mean);
  if (dens_var) std::copy(m->h_dens_var.begin(), m->h_dens_var.end(), dens_var);
  return SR_OK;
  });
}

int sr_model_tying_info(const sr_model* m, uint32_t* n_mean, uint32_t* n_var) {
  return guarded(__func__, [&]() -> int {
  if (!m) return fail(SR_EINVAL, "null model handle");
  if (n_mean) *n_mean = m->n_mean;
  if (n_var) *n_var = m->n_var;
  return SR_OK;
  });
}

int sr_accumulate_corpus(sr_model* m, sr_corpus* c, const uint16_t* states, int first_pass, int max_approx, double* mean_acc,
                         double* mean_w, double* var_acc, double* var_w) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  if (!c || c->model != m) return fail(SR_EINVAL, "corpus does not belong to this model");
  const bool to_host = mean_acc || mean_w || var_acc || var_w;  // all NULL: the statistics stay on the device
  if (to_host && (!mean_acc || !mean_w || !var_acc || !var_w)) return fail(SR_EINVAL, "null output (pass all four arrays, or none)");
  const uint64_t F = c->n_frames;
  const uint32_t D = m->dim;
  c->acc_valid = false;
  if (F == 0) {  // empty result = reset_accumulators()
    if (to_host) {
      std::fill(mean_acc, mean_acc + (size_t)m->n_mean * D, 0.0);
      std::fill(mean_w, mean_w + m->n_mean, 0.0);
      std::fill(var_acc, var_acc + (size_t)m->n_var * D, 1e-4);
      std::fill(var_w, var_w + m->n_var, 0.0);
    }
    return to_host ? SR_OK : fail(SR_EINVAL, "empty corpus: nothing to keep on the device");
  }
  if (!states) return fail(SR_EINVAL, "states is null");
  const bool soft = !first_pass && !max_approx;
  std::vector<uint64_t> pair_off(F);
  uint64_t n_pairs = 0;
  for (uint64_t f = 0; f < F; f++) {
    if (states[f] >= m->n_states) return fail(SR_EINVAL, "frame %llu: state %u >= n_states", (unsigned long long)f, states[f]);
    pair_off[f] = n_pairs;
    n_pairs += soft ? (m->h_dens_off[states[f] + 1] - m->h_dens_off[states[f]]) : 1;
  }
  if (n_pairs >= (1ull << 31)) return fail(SR_ELIMIT, "too many (frame, density) pairs");
  if (n_pairs == 0) return fail(SR_EINVAL, "no (frame, density) pairs to accumulate");
  if ((rc = srhost::corpus_ready(c, 0, F, m->s_gmm))) return rc;
  HIP_TRY(c->out_states.upload(states, F));
  HIP_TRY(c->pair_off.upload(pair_off.data(), F));
  HIP_TRY(c->pair_frame.ensure(n_pairs)); HIP_TRY(c->key_mean.ensure(n_pairs)); HIP_TRY(c->key_var.ensure(n_pairs));
  HIP_TRY(c->pair_w.ensure(n_pairs)); HIP_TRY(c->keys_sorted.ensure(n_pairs)); HIP_TRY(c->pairs_sorted.ensure(n_pairs));
  {
    std::vector<uint32_t> iota(n_pairs);
    std::iota(iota.begin(), iota.end(), 0u);
    HIP_TRY(c->iota.upload(iota.data(), n_pairs));
  }
  const size_t temp = em_sort_temp_bytes(n_pairs);
  HIP_TRY(c->sort_temp.ensure(temp));
  HIP_TRY(c->row_begin.ensure((size_t)std::max(m->n_mean, m->n_var) + 1));
  HIP_TRY(c->acc_mean.ensure((size_t)m->n_mean * D)); HIP_TRY(c->w_mean.ensure(m->n_mean));
  HIP_TRY(c->acc_var.ensure((size_t)m->n_var * D)); HIP_TRY(c->w_var.ensure(m->n_var));
  EmArgs a{};
  a.feats = c->feats.p; a.n_frames = F; a.n_pairs = n_pairs; a.dim = D; a.states = c->out_states.p; a.pair_off = c->pair_off.p;
  a.dens_off = m->dens_off.p; a.means = m->means.p; a.inv_vars = m->inv_vars.p; a.norm = m->norm.p; a.logw = m->logw.p;
  a.dens_mean = m->dens_mean.p; a.dens_var = m->dens_var.p; a.n_mean = m->n_mean; a.n_var = m->n_var;
  a.first_pass = first_pass; a.max_approx = max_approx;
  a.pair_frame = c->pair_frame.p; a.pair_w = c->pair_w.p; a.key_mean = c->key_mean.p; a.key_var = c->key_var.p;
  EventPair ep{};
  if ((rc = prof_begin(m, m->s_gmm, 1, &ep))) return rc;
  HIP_TRY(launch_em_accumulate(a, c->sort_temp.p, temp, c->iota.p, c->keys_sorted.p, c->pairs_sorted.p, c->row_begin.p, c->acc_mean.p, c->w_mean.p,
                               c->acc_var.p, c->w_var.p, m->s_gmm));
  if ((rc = prof_end(m, m->s_gmm, &ep))) return rc;
  HIP_TRY(hipStreamSynchronize(m->s_gmm));
  c->acc_valid = true; c->acc_n_mean = m->n_mean; c->acc_n_var = m->n_var;
  if (to_host) {
    HIP_TRY(hipMemcpy(mean_acc, c->acc_mean.p, sizeof(double) * (size_t)m->n_mean * D, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mean_w, c->w_mean.p, sizeof(double) * m->n_mean, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(var_acc, c->acc_var.p, sizeof(double) * (size_t)m->n_var * D, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(var_w, c->w_var.p, sizeof(double) * m->n_var, hipMemcpyDeviceToHost));
  }
  return SR_OK;
  });
}

int sr_model_create_from_accumulated(sr_model* m, sr_corpus* c, int pooling, int max_approx, sr_model** out) {
  return guarded(__func__, [&]() -> int {
  if (!out) return fail(SR_EINVAL, "out is null");
  *out = nullptr;
  int rc = check_model(m);
  if (rc) return rc;
  if (!c || c->model != m) return fail(SR_EINVAL, "corpus does not belong to this model");
  if (pooling < 0 || pooling > 2) return fail(SR_EINVAL, "pooling must be 0 (global), 1 (mixture) or 2 (none)");
  return srhost::finalize_accumulated(m, c, pooling, max_approx, out);
  });
}

int sr_probe_fp16_accumulation(int device, int* within_model, double* worst_ratio) {
  return guarded(__func__, [&]() -> int {
  if (!within_model) return fail(SR_EINVAL, "within_model is null");
  HIP_TRY(hipSetDevice(device));
  bool ok = false, ok4 = false;
  double worst = 0.0, worst4 = 0.0;
  HIP_TRY(probe_fp16_accumulation(nullptr, 3, &ok, &worst));    // K = 96: models of dimension <= 46
  HIP_TRY(probe_fp16_accumulation(nullptr, 4, &ok4, &worst4));  // K = 128: dimension 47 .. 62 (allowed: 132)
  *within_model = (ok && ok4) ? 1 : 0;
  if (worst_ratio) *worst_ratio = std::max(worst, worst4 * (87.0 / 132.0));  // on the K = 96 scale
  return SR_OK;
  });
}

int sr_probe_fp16_denormals(int device, int* preserved) {
  return guarded(__func__, [&]() -> int {
  if (!preserved) return fail(SR_EINVAL, "preserved is null");
  HIP_TRY(hipSetDevice(device));
  bool ok = false;
  HIP_TRY(probe_fp16_denormals(nullptr, &ok));
  *preserved = ok ? 1 : 0;
  return SR_OK;
  });
}

int sr_profile_enable(sr_model* m, int on) {
  return guarded(__func__, [&]() -> int {
  if (!m) return fail(SR_EINVAL, "null model handle");
  m->profiling = on != 0;
  return SR_OK;
  });
}

int sr_profile_reset(sr_model* m) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  for (auto& ep : m->events) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  m->events.clear();
  m->prof = sr_profile{};
  HIP_TRY(m->pf_counter.ensure(1));
  HIP_TRY(hipMemset(m->pf_counter.p, 0, sizeof(unsigned long long)));
  return SR_OK;
  });
}

int sr_profile_read(sr_model* m, sr_profile* out) {
  return guarded(__func__, [&]() -> int {
  int rc = check_model(m);
  if (rc) return rc;
  if (!out) return fail(SR_EINVAL, "out is null");
  HIP_TRY(hipDeviceSynchronize());
  for (auto& ep : m->events) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
    if (ep.kind == 0) { m->prof.gmm_ms += ms; m->prof.gmm_launches++; }
    else if (ep.kind == 1) { m->prof.search_ms += ms; m->prof.search_launches++; }
    else if (ep.kind == 2) m->prof.prefilter_ms += ms;
    else m->prof.refine_ms += ms;
    (void)hipEventDestroy(ep.a);
    (void)hipEventDestroy(ep.b);
  }
  m->events.clear();
  if (m->pf_counter.p) {
    unsigned long long n = 0;
    HIP_TRY(hipMemcpy(&n, m->pf_counter.p, sizeof(n), hipMemcpyDeviceToHost));
    m->prof.refined_densities = n;
  }
  *out = m->prof;
  return SR_OK;
  });
}

}  // extern "C"
