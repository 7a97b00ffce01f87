/*
 * sr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See sr_oracle.h.
 *
 * Restates, operation for operation, the reference hot path so that results are bit-identical
 * to the compiled reference (x86-64, SSE2, no FMA: sietill/Makefile:22).  Build with
 * -O2 -ffp-contract=off (oracle/Makefile); no -ffast-math.
 */
#include "sr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
static void set_err(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

struct orc_model {
  uint32_t dim, n_mean, n_var, n_mix, n_dens_total;
  int pooling, max_approx;
  double *mean_acc, *mean_w, *var_acc, *var_w;   /* accumulators as stored in the file */
  double *means, *vars, *vars_inv, *norm, *logw; /* derived by finalize */
  uint32_t *mix_off, *mix_mean, *mix_var;        /* flattened mixtures */
};

/* ------------------------------------------------------------------------------------------ */
/* MIXSET v2 reader: Mixtures.cpp:104-129 (read_accumulator), :748-830 (read)                  */

static int rd(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

static int read_acc_block(FILE* f, uint32_t dim, uint32_t* n_out, double** acc_out, double** w_out) {
  uint32_t n;
  if (!rd(f, &n, 4)) { set_err("Error reading size"); return 0; }
  double* acc = (double*)malloc(sizeof(double) * (size_t)n * dim + 8);
  double* w = (double*)malloc(sizeof(double) * (size_t)n + 8);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t d;
    if (!rd(f, &d, 4)) { set_err("Error reading dimension"); goto fail; }
    if (d != dim) { set_err("Invalid dimension"); goto fail; }
    if (!rd(f, acc + (size_t)i * dim, sizeof(double) * dim)) { set_err("Error reading features"); goto fail; }
    if (!rd(f, w + i, sizeof(double))) { set_err("Error reading weight"); goto fail; }
  }
  *n_out = n; *acc_out = acc; *w_out = w;
  return 1;
fail:
  free(acc); free(w);
  return 0;
}

/* calculate_variance, Mixtures.cpp:251-275: E[x^2] - mean^2, inverse, normalisation constant */
static void calc_variance(orc_model* m, uint32_t var_idx, const double* mean) {
  const uint32_t D = m->dim;
  double* v = m->vars + (size_t)var_idx * D;
  double* iv = m->vars_inv + (size_t)var_idx * D;
  const double* acc = m->var_acc + (size_t)var_idx * D;
  const double w = m->var_w[var_idx];
  for (uint32_t d = 0; d < D; d++) v[d] = acc[d] / w;
  for (uint32_t d = 0; d < D; d++) v[d] = v[d] - mean[d] * mean[d];
  for (uint32_t d = 0; d < D; d++) iv[d] = 1 / v[d];
  double nrm = D * log(2 * M_PI);
  for (uint32_t d = 0; d < D; d++) nrm = nrm + log(v[d]);
  m->norm[var_idx] = nrm / 2;
}

/* MixtureModel::finalize, Mixtures.cpp:374-461 */
static void finalize(orc_model* m) {
  const uint32_t D = m->dim;
  double total_obs = 0.0;
  double* tmp = (double*)malloc(sizeof(double) * D);
  for (uint32_t s = 0; s < m->n_mix; s++) {
    double mix_obs = 0.0;
    for (uint32_t k = m->mix_off[s]; k < m->mix_off[s + 1]; k++) {
      const uint32_t mi = m->mix_mean[k], vi = m->mix_var[k];
      mix_obs += m->mean_w[mi];
      for (uint32_t d = 0; d < D; d++)
        m->means[(size_t)mi * D + d] = m->mean_acc[(size_t)mi * D + d] / m->mean_w[mi];
      if (m->pooling == ORC_POOL_NONE) calc_variance(m, vi, m->means + (size_t)mi * D);
    }
    for (uint32_t k = m->mix_off[s]; k < m->mix_off[s + 1]; k++) {
      const uint32_t mi = m->mix_mean[k];
      const double wt = m->mean_w[mi] / mix_obs;
      m->logw[mi] = log(wt);
    }
    if (m->pooling == ORC_POOL_MIXTURE && m->mix_off[s + 1] > m->mix_off[s]) {
      for (uint32_t d = 0; d < D; d++) tmp[d] = 0.0;
      for (uint32_t k = m->mix_off[s]; k < m->mix_off[s + 1]; k++)
        for (uint32_t d = 0; d < D; d++) tmp[d] = tmp[d] + m->mean_acc[(size_t)m->mix_mean[k] * D + d];
      for (uint32_t d = 0; d < D; d++) tmp[d] = tmp[d] / mix_obs;
      calc_variance(m, m->mix_var[m->mix_off[s]], tmp);
    }
    total_obs += mix_obs;
  }
  if (m->pooling == ORC_POOL_GLOBAL) {
    for (uint32_t d = 0; d < D; d++) tmp[d] = 0.0;
    for (uint32_t s = 0; s < m->n_mix; s++)
      for (uint32_t k = m->mix_off[s]; k < m->mix_off[s + 1]; k++)
        for (uint32_t d = 0; d < D; d++) tmp[d] = tmp[d] + m->mean_acc[(size_t)m->mix_mean[k] * D + d];
    for (uint32_t d = 0; d < D; d++) tmp[d] = tmp[d] / total_obs;
    calc_variance(m, 0, tmp);
  }
  free(tmp);
}

orc_model* orc_model_load(const char* path, uint32_t dim, int pooling, int max_approx) {
  FILE* f = fopen(path, "rb");
  if (!f) { set_err("cannot open model file"); return NULL; }
  orc_model* m = (orc_model*)calloc(1, sizeof *m);
  m->dim = dim; m->pooling = pooling; m->max_approx = max_approx;
  uint32_t* dens_mean = NULL; uint32_t* dens_var = NULL;
  char magic[8]; uint32_t version, fdim, n_dens, n_mix;
  static const char want[8] = {'M', 'I', 'X', 'S', 'E', 'T', 0, 0};
  if (!rd(f, magic, 8)) { set_err("Error reading magic header"); goto fail; }
  if (memcmp(magic, want, 8) != 0) { set_err("Invalid magic header"); goto fail; }
  if (!rd(f, &version, 4)) { set_err("Error reading version"); goto fail; }
  if (version != 2u) { set_err("Invalid version"); goto fail; }
  if (!rd(f, &fdim, 4)) { set_err("Error reading dimension"); goto fail; }
  if (fdim != dim) { set_err("Invalid dimension"); goto fail; }
  if (!read_acc_block(f, dim, &m->n_mean, &m->mean_acc, &m->mean_w)) goto fail;
  if (!read_acc_block(f, dim, &m->n_var, &m->var_acc, &m->var_w)) goto fail;
  if (!rd(f, &n_dens, 4)) { set_err("Error reading density count"); goto fail; }
  dens_mean = (uint32_t*)malloc(4 * (size_t)n_dens + 4);
  dens_var = (uint32_t*)malloc(4 * (size_t)n_dens + 4);
  for (uint32_t i = 0; i < n_dens; i++) {
    if (!rd(f, dens_mean + i, 4)) { set_err("Error reading mean_idx"); goto fail; }
    if (dens_mean[i] >= m->n_mean) { set_err("Invalid mean_idx"); goto fail; }
    if (!rd(f, dens_var + i, 4)) { set_err("Error reading var_idx"); goto fail; }
    if (dens_var[i] >= m->n_var) { set_err("Invalid var_idx"); goto fail; }
  }
  if (!rd(f, &n_mix, 4)) { set_err("Error reading mixture count"); goto fail; }
  m->n_mix = n_mix;
  m->mix_off = (uint32_t*)calloc((size_t)n_mix + 1, 4);
  {
    size_t cap = n_dens ? n_dens : 1, cnt = 0;
    m->mix_mean = (uint32_t*)malloc(4 * cap);
    m->mix_var = (uint32_t*)malloc(4 * cap);
    for (uint32_t s = 0; s < n_mix; s++) {
      uint32_t nd;
      if (!rd(f, &nd, 4)) { set_err("Error reading density count for mixture"); goto fail; }
      for (uint32_t d = 0; d < nd; d++) {
        uint32_t di; double wt;
        if (!rd(f, &di, 4)) { set_err("Error reading density idx"); goto fail; }
        if (di >= n_dens) { set_err("Invalid density idx"); goto fail; }
        if (!rd(f, &wt, 8)) { set_err("Error reading density weight"); goto fail; }
        if (wt != m->mean_w[dens_mean[di]]) { set_err("Inconsistent density weight"); goto fail; }
        if (cnt == cap) {
          cap *= 2;
          m->mix_mean = (uint32_t*)realloc(m->mix_mean, 4 * cap);
          m->mix_var = (uint32_t*)realloc(m->mix_var, 4 * cap);
        }
        m->mix_mean[cnt] = dens_mean[di];
        m->mix_var[cnt] = dens_var[di];
        cnt++;
      }
      m->mix_off[s + 1] = (uint32_t)cnt;
    }
    m->n_dens_total = (uint32_t)cnt;
  }
  free(dens_mean); free(dens_var); dens_mean = dens_var = NULL;
  fclose(f); f = NULL;
  m->means = (double*)calloc((size_t)m->n_mean * dim + 1, 8);
  m->logw = (double*)calloc((size_t)m->n_mean + 1, 8);
  m->vars = (double*)calloc((size_t)m->n_var * dim + 1, 8);
  m->vars_inv = (double*)calloc((size_t)m->n_var * dim + 1, 8);
  m->norm = (double*)calloc((size_t)m->n_var + 1, 8);
  finalize(m);
  return m;
fail:
  if (f) fclose(f);
  free(dens_mean); free(dens_var);
  orc_model_free(m);
  return NULL;
}

void orc_model_free(orc_model* m) {
  if (!m) return;
  free(m->mean_acc); free(m->mean_w); free(m->var_acc); free(m->var_w);
  free(m->means); free(m->vars); free(m->vars_inv); free(m->norm); free(m->logw);
  free(m->mix_off); free(m->mix_mean); free(m->mix_var);
  free(m);
}

uint32_t orc_model_dim(const orc_model* m) { return m->dim; }
uint32_t orc_model_num_states(const orc_model* m) { return m->n_mix; }
uint32_t orc_model_num_means(const orc_model* m) { return m->n_mean; }
uint32_t orc_model_num_vars(const orc_model* m) { return m->n_var; }
uint32_t orc_model_num_densities(const orc_model* m) { return m->n_dens_total; }
const double* orc_model_means(const orc_model* m) { return m->means; }
const double* orc_model_vars_inv(const orc_model* m) { return m->vars_inv; }
const double* orc_model_norm(const orc_model* m) { return m->norm; }
const double* orc_model_logw(const orc_model* m) { return m->logw; }
const uint32_t* orc_model_mix_offsets(const orc_model* m) { return m->mix_off; }
const uint32_t* orc_model_mix_mean_idx(const orc_model* m) { return m->mix_mean; }
const uint32_t* orc_model_mix_var_idx(const orc_model* m) { return m->mix_var; }

/* ------------------------------------------------------------------------------------------ */
/* density_score_sse, Mixtures.cpp:645-690.  Two partial sums over even/odd dims (the two SSE
 * lanes), l0 + l1 (:633-639), scalar tail for odd D (:672-675), norm + dist/2, then -logw.   */
static inline double density_score(const orc_model* m, const float* x, uint32_t mi, uint32_t vi) {
  const uint32_t D = m->dim;
  const double* mu = m->means + (size_t)mi * D;
  const double* iv = m->vars_inv + (size_t)vi * D;
  double l0 = 0.0, l1 = 0.0;
  const uint32_t D2 = D - D % 2;
  for (uint32_t d = 0; d < D2; d += 2) {
    double a = (double)x[d] - mu[d];
    a = a * a;
    a = a * iv[d];
    l0 = l0 + a;
    double b = (double)x[d + 1] - mu[d + 1];
    b = b * b;
    b = b * iv[d + 1];
    l1 = l1 + b;
  }
  double dist = l0 + l1;
  if (D % 2 == 1) {
    double c = (double)x[D - 1] - mu[D - 1];
    dist += c * c * iv[D - 1];
  }
  double score = m->norm[vi] + dist / 2;
  score -= m->logw[mi];
  return score;
}

/* min_score (Mixtures.cpp:696-713) whatever the model's max_approx flag */
static double orc_score_argmin_raw(const orc_model* m, const float* x, uint32_t state, uint32_t* density) {
  const uint32_t b = m->mix_off[state], e = m->mix_off[state + 1];
  double best = 1e10; uint32_t bi = 0;
  for (uint32_t k = b; k < e; k++) {
    double s = density_score(m, x, m->mix_mean[k], m->mix_var[k]);
    if (s < best) { bi = k - b; best = s; }
  }
  if (density) *density = bi;
  return best;
}

double orc_score_argmin(const orc_model* m, const float* x, uint32_t state, uint32_t* density) {
  const uint32_t b = m->mix_off[state], e = m->mix_off[state + 1];
  if (m->max_approx) { /* min_score, Mixtures.cpp:696-713: seed 1e10, idx 0, strict < */
    double best = 1e10; uint32_t bi = 0;
    for (uint32_t k = b; k < e; k++) {
      double s = density_score(m, x, m->mix_mean[k], m->mix_var[k]);
      if (s < best) { bi = k - b; best = s; }
    }
    if (density) *density = bi;
    return best;
  }
  /* sum_score, Mixtures.cpp:719-728: naive -log(sum exp(-s)) accumulated left to right */
  double acc = 0.0;
  for (uint32_t k = b; k < e; k++) acc += exp(-1 * density_score(m, x, m->mix_mean[k], m->mix_var[k]));
  if (density) *density = 0;
  return -1 * log(acc);
}

double orc_score(const orc_model* m, const float* x, uint32_t state) { return orc_score_argmin(m, x, state, NULL); }

void orc_score_matrix(const orc_model* m, const float* feats, size_t T, double* out, int n_threads) {
  const size_t S = m->n_mix, D = m->dim;
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 1 ? n_threads : 1)
#endif
  for (long t = 0; t < (long)T; t++)
    for (size_t s = 0; s < S; s++) out[(size_t)t * S + s] = orc_score(m, feats + (size_t)t * D, (uint32_t)s);
}

/* ------------------------------------------------------------------------------------------ */
void orc_accumulate(const orc_model* m, const float* feats, size_t T, const uint16_t* states, int first_pass,
                    int max_approx, double* mean_acc, double* mean_w, double* var_acc, double* var_w) {
  const uint32_t D = m->dim;
  for (size_t i = 0; i < (size_t)m->n_mean * D; i++) mean_acc[i] = 0.0;   /* reset_accumulators, :235-247 */
  for (size_t i = 0; i < m->n_mean; i++) mean_w[i] = 0.0;
  for (size_t i = 0; i < (size_t)m->n_var * D; i++) var_acc[i] = 1e-4;    /* minimal_variance_value_, :167,243 */
  for (size_t i = 0; i < m->n_var; i++) var_w[i] = 0.0;
  double* p = (double*)malloc(sizeof(double) * (m->n_dens_total ? m->n_dens_total : 1));
  for (size_t t = 0; t < T; t++) {
    const float* x = feats + t * D;
    const uint32_t s = states[t], b = m->mix_off[s], n = m->mix_off[s + 1] - b;
    for (uint32_t d = 0; d < n; d++) p[d] = 0.0;
    uint32_t arg = 0;
    if (max_approx && !first_pass) (void)orc_score_argmin_raw(m, x, s, &arg);
    for (uint32_t d = 0; d < n; d++) {
      if (first_pass) p[0] = 1.0;
      else if (max_approx && d == arg) p[d] = 1.0;
      else if (!max_approx) p[d] = exp(-1 * density_score(m, x, m->mix_mean[b + d], m->mix_var[b + d]));
    }
    if (!max_approx) {
      double sum = 0.0;
      for (uint32_t d = 0; d < n; d++) sum += p[d];
      for (uint32_t d = 0; d < n; d++) p[d] = p[d] / sum;
    }
    for (uint32_t d = 0; d < n; d++) {
      if (p[d] < 1e-8) continue;
      const uint32_t mi = m->mix_mean[b + d], vi = m->mix_var[b + d];
      mean_w[mi] += p[d];
      var_w[vi] += p[d];
      for (uint32_t k = 0; k < D; k++) mean_acc[(size_t)mi * D + k] = mean_acc[(size_t)mi * D + k] + p[d] * x[k];
      for (uint32_t k = 0; k < D; k++) var_acc[(size_t)vi * D + k] = var_acc[(size_t)vi * D + k] + p[d] * x[k] * x[k];
    }
  }
  free(p);
}

/* ------------------------------------------------------------------------------------------ */
double orc_tdp_score(const orc_tdp* tdp, uint16_t to, size_t jump) { /* TdpModel.cpp:19-29 */
  if (to == tdp->silence_state) return tdp->forward;
  switch (jump) {
    case 0: return tdp->loop;
    case 1: return tdp->forward;
    case 2: return tdp->skip;
  }
  return INFINITY;
}

/* emission provider: lazy per-frame cache like Recognizer.cpp:123,148-151 or a dense table */
typedef struct {
  const orc_model* m; const double* dense; size_t stride; uint32_t dim;
  const float* feats; double* cache; size_t n_cache; uint64_t n_scored;
} emis;

static inline double emis_get(emis* e, size_t t, uint16_t s) {
  if (e->dense) return e->dense[t * e->stride + s];
  if (e->cache[s] == INFINITY) {
    e->cache[s] = orc_score(e->m, e->feats + t * e->dim, s);
    e->n_scored++;
  }
  return e->cache[s];
}
static inline void emis_next_frame(emis* e) {
  if (!e->dense) for (size_t i = 0; i < e->n_cache; i++) e->cache[i] = INFINITY;
}

/* Book, Recognizer.hpp:75-89 */
typedef struct { double score; uint16_t word; uint16_t bkp; int pos; } hyp_t;

size_t orc_decode_pruned(const orc_model* m, const double* dense, size_t dense_stride,
                         const orc_lexicon* lex, const orc_tdp* tdp, const orc_search_params* sp,
                         const float* feats, size_t T, uint32_t dim, uint32_t* out_words,
                         double* tb_score, uint16_t* tb_word, uint16_t* tb_bkp, uint64_t* n_scored) {
  const size_t W = lex->n_words, S = lex->n_states;
  size_t max_pos = 0;
  size_t* n_pos = (size_t*)malloc(sizeof(size_t) * W);
  for (size_t w = 0; w < W; w++) {
    n_pos[w] = lex->word_off[w + 1] - lex->word_off[w];
    if (n_pos[w] > max_pos) max_pos = n_pos[w];
  }
  /* Recognizer.cpp:116-117 allocates n_states * max_pos hypotheses and indexes them by word * max_pos + pos: enough because
   * Lexicon::add_word gives every word fresh states (n_states >= n_words).  A lexicon whose words SHARE states can have fewer
   * states than words; the reference would index past its arrays there (undefined).  The restatement allocates for the words
   * it has -- the defined extension, and what the GPU kernels compute. */
  const size_t n_slots = (S > W ? S : W) * max_pos;
  const hyp_t dead = {INFINITY, 0, 0, 0};
  hyp_t* cur = (hyp_t*)malloc(sizeof(hyp_t) * n_slots);
  hyp_t* nxt = (hyp_t*)malloc(sizeof(hyp_t) * n_slots);
  for (size_t i = 0; i < n_slots; i++) cur[i] = nxt[i] = dead;
  hyp_t* tb = (hyp_t*)malloc(sizeof(hyp_t) * (T + 1));
  for (size_t i = 0; i <= T; i++) { tb[i].score = 0.0; tb[i].word = 0; tb[i].bkp = 0; tb[i].pos = 0; }
  emis e = {m, dense, dense_stride, dim, feats, NULL, S, 0};
  if (!dense) { e.cache = (double*)malloc(sizeof(double) * S); emis_next_frame(&e); }

  cur[0].score = 0.0; /* :120 */
  int t = 1;
  for (size_t f = 0; f < T; f++, t++) {
    double best = INFINITY;
    for (size_t i = 0; i < n_slots; i++) {
      const hyp_t* h = &cur[i];
      if (h->score == INFINITY) continue;
      if ((size_t)h->pos == n_pos[h->word] - 1) {
        /* word boundary: start every word at position 0 or 1 (:133-158) */
        for (size_t w = 0; w < W; w++) {
          const double wp = (w != lex->silence_idx) ? sp->word_penalty : 0.0;
          const uint16_t first = lex->automaton[lex->word_off[w]];
          for (size_t init = 0; init <= 1; init++) {
            hyp_t* tg = &nxt[max_pos * w + init];
            double ns = h->score + wp + orc_tdp_score(tdp, first, init + 1);
            if (ns > tg->score) continue;
            ns += emis_get(&e, f, first);
            if (tg->score > ns) {
              tg->score = ns; tg->pos = (int)init; tg->bkp = (uint16_t)(t - 1); tg->word = (uint16_t)w;
              best = ns < best ? ns : best;
            }
          }
        }
      } else {
        /* within-word 0-1-2 expansion (:162-187) */
        for (size_t jump = 0; jump <= 2; jump++) {
          const uint16_t np = (uint16_t)(h->pos + jump);
          if ((size_t)np >= n_pos[h->word]) break;
          const uint16_t st = lex->automaton[lex->word_off[h->word] + np];
          hyp_t* tg = &nxt[max_pos * h->word + np];
          double ns = h->score + orc_tdp_score(tdp, st, jump);
          if (ns > tg->score) continue;
          ns += emis_get(&e, f, st);
          if (tg->score > ns) {
            tg->score = ns; tg->pos = np; tg->bkp = h->bkp; tg->word = h->word;
            best = ns < best ? ns : best;
          }
        }
      }
    }
    tb[t].score = INFINITY; /* :191 */
    for (size_t i = 0; i < n_slots; i++) {
      hyp_t* h = &nxt[i];
      if (h->score > best + sp->am_threshold) { h->score = INFINITY; continue; }
      if ((size_t)h->pos == n_pos[h->word] - 1) {
        if (tb[t].score > h->score) { tb[t].score = h->score; tb[t].word = h->word; tb[t].bkp = h->bkp; }
      }
    }
    hyp_t* sw = cur; cur = nxt; nxt = sw;
    for (size_t i = 0; i < n_slots; i++) nxt[i] = dead;
    emis_next_frame(&e);
  }
  /* traceback (:222-231) */
  size_t n_out = 0;
  t = (int)T;
  while (t > 0) {
    if (tb[t].word != lex->silence_idx) out_words[n_out++] = tb[t].word;
    t = tb[t].bkp;
  }
  for (size_t i = 0; i < n_out / 2; i++) { uint32_t x = out_words[i]; out_words[i] = out_words[n_out - 1 - i]; out_words[n_out - 1 - i] = x; }
  for (size_t i = 0; i <= T; i++) {
    if (tb_score) tb_score[i] = tb[i].score;
    if (tb_word) tb_word[i] = tb[i].word;
    if (tb_bkp) tb_bkp[i] = tb[i].bkp;
  }
  if (n_scored) *n_scored = e.n_scored;
  free(n_pos); free(cur); free(nxt); free(tb); free(e.cache);
  return n_out;
}

/* ------------------------------------------------------------------------------------------ */
double orc_align_full(const orc_model* m, const double* dense, size_t dense_stride, const orc_tdp* tdp,
                      const uint16_t* ref, size_t N, const float* feats, size_t T, uint32_t dim,
                      uint16_t* out_states) {
  if (N < 1 || T < 1 || N > T) { set_err("align_full requires 1 <= N <= T"); return NAN; }
  const int Ti = (int)T, Ni = (int)N;
  int* bp = (int*)malloc(sizeof(int) * N * T); /* [state][frame], Alignment.cpp:54-60 */
  for (size_t i = 0; i < N * T; i++) bp[i] = -1;
  double* prev = (double*)malloc(sizeof(double) * T);
  double* cur = (double*)malloc(sizeof(double) * T);
  for (size_t i = 0; i < T; i++) prev[i] = cur[i] = INFINITY;
  /* the aligner scores without a cache (the check at :85-91 never hits) */
#define AM(t, s) (dense ? dense[(size_t)(t) * dense_stride + (s)] : orc_score(m, feats + (size_t)(t) * dim, (s)))
  int max_state = 2, min_state = Ni - 1 - (Ti - 2) * 2; /* :73-74 */
  prev[0] = AM(0, ref[0]);                              /* :77 */
  for (int t = 1; t < Ti; t++, min_state += 2, max_state += 2) {
    const int lo = min_state > 0 ? min_state : 0, hi = (Ni - 1 < max_state) ? Ni - 1 : max_state;
    for (int s = lo; s <= hi; s++) {
      const double local = AM(t, ref[s]);
      double best = prev[s] + orc_tdp_score(tdp, ref[s], 0);
      int taken = 0;
      if (s > 0) {
        const double fw = prev[s - 1] + orc_tdp_score(tdp, ref[s - 1], 1);
        if (fw < best) { best = fw; taken = 1; }
      }
      if (s > 1) {
        const double sk = prev[s - 2] + orc_tdp_score(tdp, ref[s - 2], 2);
        if (sk < best) { best = sk; taken = 2; }
      }
      cur[s] = local + best;
      bp[(size_t)s * T + t] = s - taken;
    }
    memcpy(prev, cur, sizeof(double) * T); /* :124 */
  }
#undef AM
  size_t si = N - 1;
  for (long t = (long)T - 1; t >= 0; t--) { /* :129-138 */
    out_states[t] = ref[si];
    si = (size_t)bp[si * T + (size_t)t];
  }
  const double cost = cur[N - 1];
  free(bp); free(prev); free(cur);
  return cost;
}

double orc_align_pruned(const orc_model* m, const double* dense, size_t dense_stride, const orc_tdp* tdp,
                        const uint16_t* ref, size_t N, const float* feats, size_t T, uint32_t dim,
                        double thr, uint16_t* out_states) {
  if (N < 1 || T < 1) { set_err("align_pruned requires N,T >= 1"); return NAN; }
  /* beams as dense arrays: score[t][pos] (+inf = absent), pred[t][pos] */
  double* sc = (double*)malloc(sizeof(double) * N * T);
  int32_t* pred = (int32_t*)malloc(sizeof(int32_t) * N * T);
  uint8_t* alive = (uint8_t*)calloc(N * T, 1);
  double* cache = (double*)malloc(sizeof(double) * N);
#define AMC(t, p) (dense ? dense[(size_t)(t) * dense_stride + ref[p]] \
                         : (cache[p] == INFINITY ? (cache[p] = orc_score(m, feats + (size_t)(t) * dim, ref[p])) : cache[p]))
  sc[0] = dense ? dense[ref[0]] : orc_score(m, feats, ref[0]); /* :158 */
  alive[0] = 1; pred[0] = -1;
  for (size_t t = 1; t < T; t++) {
    for (size_t p = 0; p < N; p++) cache[p] = INFINITY;
    double* ns = sc + t * N; int32_t* np = pred + t * N; uint8_t* na = alive + t * N;
    const double* ps = sc + (t - 1) * N; const uint8_t* pa = alive + (t - 1) * N;
    for (size_t p = 0; p < N; p++) {
      if (!pa[p]) continue;
      for (size_t jump = 0; jump <= 2; jump++) {
        const uint16_t q = (uint16_t)(p + jump);
        if (q >= N) break;
        double c = ps[p];
        c += orc_tdp_score(tdp, ref[q], jump); /* keyed on destination, :191 */
        c += AMC(t, q);
        if (!na[q]) { na[q] = 1; ns[q] = c; np[q] = (int32_t)p; }
        else if (ns[q] > c) { ns[q] = c; np[q] = (int32_t)p; }
      }
    }
    double best = INFINITY;
    for (size_t p = 0; p < N; p++) if (na[p] && best > ns[p]) best = ns[p];
    const double ub = best + thr;
    for (size_t p = 0; p < N; p++) if (na[p] && ns[p] > ub) na[p] = 0;
  }
#undef AMC
  size_t hi = 0;
  for (size_t p = 0; p < N; p++) if (alive[(T - 1) * N + p]) hi = p; /* highest reached, :244-251 */
  const double cost = sc[(T - 1) * N + hi];
  size_t p = hi;
  for (size_t t = T - 1; t > 0; t--) { /* :258-270 */
    out_states[t] = ref[p];
    p = (size_t)pred[t * N + p];
  }
  out_states[0] = ref[0]; /* :273-276 */
  free(sc); free(pred); free(alive); free(cache);
  return cost;
}

/* ------------------------------------------------------------------------------------------ */
typedef struct { uint16_t total, sub, ins, del; } ed_t;

void orc_edit_distance(const uint64_t* ref, size_t n_ref, const uint64_t* hyp, size_t n_hyp, uint16_t out4[4]) {
  ed_t* cur = (ed_t*)calloc(n_ref + 1, sizeof(ed_t));
  ed_t* prev = (ed_t*)calloc(n_ref + 1, sizeof(ed_t));
  for (size_t r = 1; r <= n_ref; r++) { cur[r] = cur[r - 1]; cur[r].total++; cur[r].del++; }
  for (size_t h = 1; h <= n_hyp; h++) {
    ed_t* sw = cur; cur = prev; prev = sw; /* :349 */
    cur[0].total++; cur[0].ins++;          /* :352 (operates on the stale row, as the reference does) */
    for (size_t r = 1; r <= n_ref; r++) {
      uint16_t best = 0xFFFF;
      if (prev[r - 1].total < best && ref[r - 1] == hyp[h - 1]) { cur[r] = prev[r - 1]; best = cur[r].total; }
      if (prev[r - 1].total + 1 < best) { cur[r] = prev[r - 1]; cur[r].total++; cur[r].sub++; best = cur[r].total; }
      if (prev[r].total + 1 < best) { cur[r] = prev[r]; cur[r].total++; cur[r].ins++; best = cur[r].total; }
      if (cur[r - 1].total + 1 < best) { cur[r] = cur[r - 1]; cur[r].total++; cur[r].del++; best = cur[r].total; }
    }
  }
  out4[0] = cur[n_ref].total; out4[1] = cur[n_ref].sub; out4[2] = cur[n_ref].ins; out4[3] = cur[n_ref].del;
  free(cur); free(prev);
}

/* ------------------------------------------------------------------------------------------ */
static double now_s(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_recognize_batch(const orc_model* m, const orc_lexicon* lex, const orc_tdp* tdp,
                           const orc_search_params* sp, const float* feats, const uint64_t* frame_off,
                           size_t n_utts, uint32_t dim, int n_threads, uint32_t* out_words,
                           uint64_t* out_word_off) {
  size_t* counts = (size_t*)calloc(n_utts + 1, sizeof(size_t));
  (void)n_threads;
  const double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(n_threads > 1 ? n_threads : 1)
#endif
  for (long u = 0; u < (long)n_utts; u++) {
    const size_t b = frame_off[u], T = frame_off[u + 1] - frame_off[u];
    /* every utterance writes into its own [b, b+T) window; compacted below */
    counts[u] = orc_decode_pruned(m, NULL, 0, lex, tdp, sp, feats + b * dim, T, dim, out_words + b,
                                  NULL, NULL, NULL, NULL);
  }
  const double t1 = now_s();
  size_t w = 0;
  out_word_off[0] = 0;
  for (size_t u = 0; u < n_utts; u++) {
    const size_t b = frame_off[u];
    memmove(out_words + w, out_words + b, counts[u] * sizeof(uint32_t));
    w += counts[u];
    out_word_off[u + 1] = w;
  }
  free(counts);
  return t1 - t0;
}

/* ---- B1: bigram linear-lexicon beam search ---------------------------------------------------------------------
 * Restates Teaching::LinearSearch (rwth-asr-0.5/src/Teaching/LinearSearch.cc).  Containers become growing arrays;
 * BookKeeping's mark-and-sweep (BookKeeping.cc:23-47, every 50 frames) only recycles entries no live hypothesis can
 * reach, so an append-only book yields the same traceback items (indices differ, contents do not). */
#include <float.h>

typedef struct { uint32_t word; float score; uint32_t bp; } bg_wb;                 /* WordBoundaryHypothesis :27-43 (.hh) */
typedef struct { uint16_t state; float score; uint32_t bp; } bg_sh;                /* StateHypothesis :10-21 */
typedef struct { uint32_t word, begin, end, entry; } bg_wh;                        /* WordHypothesis :23-38 */
typedef struct { uint32_t word; float score; uint32_t bp, time; } bg_entry;        /* BookKeeping::Entry */
#define BG_INVALID 0xFFFFFFFFu

typedef struct {
  uint32_t W, silence;
  const uint32_t* word_off; const uint16_t* mixtures;
  bg_sh *sh, *nsh; size_t n_sh, n_nsh, cap_sh;
  bg_wh* wh; size_t n_wh, cap_wh;
  uint32_t* sh_map;   /* stateHypothesisMap_ [maxlen + 1] */
  uint32_t* wh_map;   /* wordHypothesisMap_ [2W] */
  bg_entry* book; size_t n_book, cap_book;
  float tdp[2][4];
  uint32_t first_new;
  float best;
} bg_space;

static uint32_t bg_map_copy(const bg_space* s, uint32_t w) { return w == s->silence ? w : w % s->W; }      /* :197-200 */
static uint32_t bg_sil_copy(const bg_space* s, uint32_t w) { return w == s->silence ? w : w + s->W; }      /* :202-205 */
static uint32_t bg_ac_word(const bg_space* s, uint32_t w) { return w < s->W ? w : s->silence; }           /* :207-210 */
static int bg_is_sil(const bg_space* s, uint32_t w) { return w == s->silence || w >= s->W; }              /* :212-215 */
static uint32_t bg_len(const bg_space* s, uint32_t w) { uint32_t a = bg_ac_word(s, w); return s->word_off[a + 1] - s->word_off[a]; }

static void bg_push_sh(bg_sh** v, size_t* n, size_t* cap, bg_sh h) {
  if (*n == *cap) { *cap = *cap ? 2 * *cap : 1024; *v = (bg_sh*)realloc(*v, *cap * sizeof(bg_sh)); }
  (*v)[(*n)++] = h;
}

static uint32_t bg_add_entry(bg_space* s, uint32_t word, float score, uint32_t bp, uint32_t t) {  /* BookKeeping::addEntry */
  if (s->n_book == s->cap_book) { s->cap_book *= 2; s->book = (bg_entry*)realloc(s->book, s->cap_book * sizeof(bg_entry)); }
  bg_entry e = {word, score, bp, t};
  s->book[s->n_book] = e;
  return (uint32_t)s->n_book++;
}

/* addBookKeepingEntries :397-418 (tagActiveEntries omitted, see above) */
static void bg_book_keeping(bg_space* s, uint32_t t, bg_wb* we, size_t n_we) {
  for (size_t i = 0; i < n_we; i++) {
    const uint32_t nb = bg_add_entry(s, bg_ac_word(s, we[i].word), we[i].score, we[i].bp, t);
    we[i].bp = nb;
    if (t == 0) s->book[nb].bp = nb;  /* self loop */
    if (we[i].word == s->silence) {   /* avoid chains of silence */
      bg_entry* cur = &s->book[nb];
      if (s->book[cur->bp].word == s->silence) cur->bp = s->book[cur->bp].bp;
    }
  }
}

/* expandState :296-326 */
static void bg_expand_state(bg_space* s, uint32_t word, bg_sh h, size_t* cap_nsh) {
  const uint32_t n = bg_len(s, word);
  uint32_t hi = (uint32_t)h.state + 2u;
  if (hi > n) hi = n;
  for (uint32_t succ = h.state > 1 ? h.state : 1; succ <= hi; succ++) {
    float ns = h.score;
    const uint32_t td = succ - h.state;
    if (h.state || td > 1) ns += s->tdp[bg_is_sil(s, word)][td];
    const uint32_t idx = s->sh_map[succ];
    if (idx < s->first_new || idx >= s->n_nsh || s->nsh[idx].state != succ) {
      s->sh_map[succ] = (uint32_t)s->n_nsh;
      bg_sh nh = {(uint16_t)succ, ns, h.bp};
      bg_push_sh(&s->nsh, &s->n_nsh, cap_nsh, nh);
    } else if (s->nsh[idx].score >= ns) {
      s->nsh[idx].score = ns;
      s->nsh[idx].bp = h.bp;
    }
  }
}

size_t orc_bigram_decode(const double* dense, size_t dense_stride, size_t T, uint32_t W, uint32_t silence,
                         const uint32_t* word_off, const uint16_t* mixtures, const float* lm, const float tdp[2][4],
                         float acoustic_pruning, float lm_pruning, uint32_t* out_word, float* out_score,
                         uint32_t* out_time, size_t cap, uint64_t* stats) {
  bg_space sp;
  memset(&sp, 0, sizeof(sp));
  sp.W = W; sp.silence = silence; sp.word_off = word_off; sp.mixtures = mixtures;
  memcpy(sp.tdp, tdp, sizeof(sp.tdp));
  uint32_t maxlen = 0;
  for (uint32_t w = 0; w < W; w++) if (word_off[w + 1] - word_off[w] > maxlen) maxlen = word_off[w + 1] - word_off[w];
  sp.sh_map = (uint32_t*)malloc((maxlen + 1) * sizeof(uint32_t));
  sp.wh_map = (uint32_t*)malloc(2 * (size_t)W * sizeof(uint32_t));
  for (uint32_t i = 0; i <= maxlen; i++) sp.sh_map[i] = BG_INVALID;   /* reset :182-192 */
  for (uint32_t i = 0; i < 2 * W; i++) sp.wh_map[i] = BG_INVALID;
  size_t cap_nsh = 0;
  sp.cap_wh = 2 * (size_t)W; sp.wh = (bg_wh*)malloc(sp.cap_wh * sizeof(bg_wh));
  sp.cap_book = 1024; sp.book = (bg_entry*)malloc(sp.cap_book * sizeof(bg_entry));
  { bg_entry e0 = {BG_INVALID, FLT_MAX, 0, 0}; sp.book[0] = e0; sp.n_book = 1; }   /* index 0: the sentinel (BookKeeping.cc:6,12) */
  sp.best = FLT_MAX;
  bg_wb* we = (bg_wb*)malloc((2 * (size_t)W + 1) * sizeof(bg_wb));
  bg_wb* ws = (bg_wb*)malloc((2 * (size_t)W + 1) * sizeof(bg_wb));
  size_t n_we = 0, n_ws = 0;
  uint32_t* first_idx = (uint32_t*)malloc((size_t)W * sizeof(uint32_t));
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;

  /* initialize: addInitialHypothesis :211-216 */
  { bg_wb w0 = {silence, 0.0f, 0}; we[n_we++] = w0; }
  bg_book_keeping(&sp, 0, we, n_we);

  for (size_t t = 1; t <= T; t++) {   /* processFrame(t), t = 1..T (SearchInterface.hh:58-60) */
    const double* row = dense + (t - 1) * dense_stride;
    /* bigramRecombination :219-244 */
    n_ws = W;
    for (uint32_t w = 0; w < W; w++) { bg_wb d = {BG_INVALID, FLT_MAX, BG_INVALID}; ws[w] = d; }
    for (size_t e = 0; e < n_we; e++) {
      const uint32_t prev = bg_map_copy(&sp, we[e].word);
      for (uint32_t w = 0; w < W; w++) {
        if (w == silence) continue;
        const float ns = we[e].score + lm[(size_t)w * W + prev];
        if (ns < ws[w].score) { bg_wb h = {w, ns, we[e].bp}; ws[w] = h; }
      }
      if (we[e].word < W) { bg_wb h = {bg_sil_copy(&sp, we[e].word), we[e].score, we[e].bp}; ws[n_ws++] = h; }
    }
    /* LM beam :498-503 (std::min_element: first minimum; only its score is used) */
    float best_start = ws[0].score;
    for (size_t i = 1; i < n_ws; i++) if (ws[i].score < best_start) best_start = ws[i].score;
    float lm_thr = lm_pruning;
    if (lm_thr < FLT_MAX) lm_thr += best_start;
    /* insertWordStartHypotheses :246-255 + addEntryStateHypothesis :257-268 */
    for (size_t i = 0; i < n_ws; i++) {
      if (!(ws[i].score < lm_thr)) continue;
      const uint32_t word = ws[i].word;
      if (sp.wh_map[word] == BG_INVALID) {
        sp.wh_map[word] = (uint32_t)sp.n_wh;
        bg_wh nh = {word, BG_INVALID, BG_INVALID, BG_INVALID};
        sp.wh[sp.n_wh++] = nh;
      }
      sp.wh[sp.wh_map[word]].entry = (uint32_t)sp.n_sh;
      bg_sh eh = {0, ws[i].score, ws[i].bp};
      bg_push_sh(&sp.sh, &sp.n_sh, &sp.cap_sh, eh);
    }
    /* expandHypotheses :270-294 */
    sp.n_nsh = 0;
    for (size_t w = 0; w < sp.n_wh; w++) {
      bg_wh* wh = &sp.wh[w];
      sp.first_new = (uint32_t)sp.n_nsh;
      if (wh->entry != BG_INVALID) { bg_expand_state(&sp, wh->word, sp.sh[wh->entry], &cap_nsh); wh->entry = BG_INVALID; }
      /* (a freshly activated word has begin = end = invalidIndex: the loop below does not run) */
      for (uint32_t i = wh->begin; i < wh->end; i++) bg_expand_state(&sp, wh->word, sp.sh[i], &cap_nsh);
      wh->begin = sp.first_new;
      wh->end = (uint32_t)sp.n_nsh;
    }
    { bg_sh* tmp = sp.sh; sp.sh = sp.nsh; sp.nsh = tmp; size_t c = sp.cap_sh; sp.cap_sh = cap_nsh; cap_nsh = c; sp.n_sh = sp.n_nsh; }
    /* addAcousticScores :328-339 */
    sp.best = FLT_MAX;
    for (size_t w = 0; w < sp.n_wh; w++) {
      const uint32_t a = bg_ac_word(&sp, sp.wh[w].word);
      for (uint32_t i = sp.wh[w].begin; i < sp.wh[w].end; i++) {
        sp.sh[i].score += (float)row[mixtures[word_off[a] + sp.sh[i].state - 1]];
        if (sp.sh[i].score < sp.best) sp.best = sp.sh[i].score;
      }
    }
    float ac_thr = acoustic_pruning;
    if (ac_thr < FLT_MAX) ac_thr += sp.best;
    /* pruneStatesAndFindWordEnds :341-376 */
    n_we = 0;
    {
      size_t in = 0, out = 0, wout = 0;
      for (size_t w = 0; w < sp.n_wh; w++) {
        bg_wh* wh = &sp.wh[w];
        const uint32_t end = wh->end;
        wh->begin = (uint32_t)out;
        const uint32_t n = bg_len(&sp, wh->word);
        for (; in < end; in++) {
          const float sc = sp.sh[in].score + sp.tdp[bg_is_sil(&sp, wh->word)][3];
          if (sc < ac_thr) {
            sp.sh[out++] = sp.sh[in];
            if (sp.sh[in].state == n) { bg_wb h = {wh->word, sc, sp.sh[in].bp}; we[n_we++] = h; }
          }
        }
        wh->end = (uint32_t)out;
        if (wh->end - wh->begin > 0) { sp.wh_map[wh->word] = (uint32_t)wout; sp.wh[wout++] = *wh; }
        else sp.wh_map[wh->word] = BG_INVALID;
      }
      sp.n_wh = wout;
      sp.n_sh = out;
    }
    /* mergeSilenceToBigramNodes :378-395 -- including its quirk: the merged hypotheses sit at the index of their
     * first occurrence, but the list is then cut to its first nWordEnds entries */
    {
      for (uint32_t w = 0; w < W; w++) first_idx[w] = BG_INVALID;
      size_t n_ends = 0;
      for (size_t e = 0; e < n_we; e++) {
        const uint32_t word = bg_map_copy(&sp, we[e].word);
        if (first_idx[word] == BG_INVALID) { first_idx[word] = (uint32_t)e; n_ends++; }
        const uint32_t wi = first_idx[word];
        if (we[e].score <= we[wi].score) we[wi] = we[e];
      }
      n_we = n_ends;
    }
    bg_book_keeping(&sp, (uint32_t)t, we, n_we);
    if (stats) { stats[0] += n_we; stats[1] += sp.n_wh; stats[2] += sp.n_sh; stats[3] = sp.n_book; }
  }

  /* traceback :420-436 */
  size_t n_out = 0;
  if (n_we > 0) {
    size_t bi = 0;
    for (size_t i = 1; i < n_we; i++) if (we[i].score < we[bi].score) bi = i;
    uint32_t bp = we[bi].bp;
    size_t len = 0;
    for (uint32_t b = bp; sp.book[b].time > 0; b = sp.book[b].bp) len++;
    n_out = len;
    for (uint32_t b = bp; sp.book[b].time > 0; b = sp.book[b].bp) {
      len--;
      if (len < cap) { out_word[len] = sp.book[b].word; out_score[len] = sp.book[b].score; out_time[len] = sp.book[b].time; }
    }
  }
  free(sp.sh); free(sp.nsh); free(sp.wh); free(sp.sh_map); free(sp.wh_map); free(sp.book); free(we); free(ws); free(first_idx);
  return n_out;
}
