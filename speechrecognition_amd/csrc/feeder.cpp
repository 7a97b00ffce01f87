// feeder.cpp -- frame-batch feeder and multi-device driver of libsrgpu.so (include/srgpu.h).
//
// The reference hands its recogniser one contiguous float buffer per corpus (Corpus::read, sietill/Corpus.cpp:89-111) and
// spreads the segments over host threads (`#pragma omp parallel for`, Recognizer.cpp:46-47).  Here:
//
//   * sr_corpus_upload_async: the buffer goes to the device in 2 MiB pieces through two PINNED staging buffers -- a feeder
//     thread fills one while the copy engine drains the other on the corpus' own stream -- and the compute entry points
//     wait, per score chunk and on the device, only for the pieces that chunk needs (srhost::corpus_ready): the transfer of
//     later utterances overlaps the scoring of earlier ones.  A piece list instead of one pointer lets the multi-device
//     driver feed a shard straight from the caller's buffer, utterance by utterance, without assembling it first.
//   * sr_shard_utterances: greedy longest-processing-time deal of utterances to devices by frame count.
//   * sr_recognize_batch_multi: one host thread per device handle feeds and recognises its shard; results are gathered
//     on the host in corpus order.  No collective anywhere: utterances are independent.
#include <algorithm>
#include <cstring>
#include <memory>
#include <numeric>

#include "handles.h"

using srhost::fail;
using srhost::guarded;

namespace {
constexpr size_t kPieceBytes = 2u << 20;  // per staging buffer; 47 MB of features = 23 pieces (8 MiB pieces made the first scoring
                                          // launch wait 1.1 ms for its data: profiles/r3_batch_boundary.txt)
struct Segment { const float* src; uint64_t n_floats; };  // host memory, copied in order to consecutive device addresses
}  // namespace

struct sr_feeder {
  std::thread thread;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Segment> segments;
  // pieces in device order: piece i covers floats [i * piece_floats, ...); issued[i] once its copy has been queued
  uint64_t total_floats = 0, piece_floats = 0;
  uint32_t n_pieces = 0, n_issued = 0;   // n_issued guarded by mu
  std::vector<hipEvent_t> done;          // recorded on s_copy behind piece i
  int rc = SR_OK;                        // guarded by mu; the feeder's first error
  std::string err;
  bool finished = false;
  hipStream_t s_copy = nullptr;
  float* staging[2] = {nullptr, nullptr};
  hipEvent_t staging_free[2] = {nullptr, nullptr};
  sr_model* lender = nullptr;  // the stream, buffers and events above are the model's (returned at the end), else our own
  int device = 0;
  float* dst = nullptr;
};

namespace {

void feeder_main(sr_feeder* fd) {
  auto stop = [&](hipError_t e, const char* what) {
    std::lock_guard<std::mutex> lk(fd->mu);
    if (fd->rc == SR_OK) { fd->rc = SR_EHIP; fd->err = std::string(what) + ": " + hipGetErrorString(e); }
    fd->n_issued = fd->n_pieces;  // nobody waits for pieces that will not come
    fd->finished = true;
    fd->cv.notify_all();
  };
  hipError_t e = hipSetDevice(fd->device);
  if (e != hipSuccess) return stop(e, "hipSetDevice (feeder)");
  size_t seg = 0;
  uint64_t seg_pos = 0;
  for (uint32_t i = 0; i < fd->n_pieces; i++) {
    const int b = (int)(i & 1);
    const uint64_t first = (uint64_t)i * fd->piece_floats;
    const uint64_t n = std::min<uint64_t>(fd->piece_floats, fd->total_floats - first);
    if (i >= 2 && (e = hipEventSynchronize(fd->staging_free[b])) != hipSuccess) return stop(e, "hipEventSynchronize (staging)");
    uint64_t filled = 0;
    while (filled < n) {  // gather from the caller's segments
      const Segment& sg = fd->segments[seg];
      const uint64_t take = std::min<uint64_t>(n - filled, sg.n_floats - seg_pos);
      memcpy(fd->staging[b] + filled, sg.src + seg_pos, take * sizeof(float));
      filled += take;
      seg_pos += take;
      if (seg_pos == sg.n_floats) { seg++; seg_pos = 0; }
    }
    if ((e = hipMemcpyAsync(fd->dst + first, fd->staging[b], n * sizeof(float), hipMemcpyHostToDevice, fd->s_copy)) != hipSuccess)
      return stop(e, "hipMemcpyAsync (feeder)");
    if ((e = hipEventRecord(fd->done[i], fd->s_copy)) != hipSuccess || (e = hipEventRecord(fd->staging_free[b], fd->s_copy)) != hipSuccess)
      return stop(e, "hipEventRecord (feeder)");
    {
      std::lock_guard<std::mutex> lk(fd->mu);
      fd->n_issued = i + 1;
    }
    fd->cv.notify_all();
  }
  e = hipStreamSynchronize(fd->s_copy);  // the caller's buffer is free again once this thread has ended
  if (e != hipSuccess) return stop(e, "hipStreamSynchronize (feeder)");
  std::lock_guard<std::mutex> lk(fd->mu);
  fd->finished = true;
  fd->cv.notify_all();
}

void feeder_free(sr_feeder* fd) {
  if (!fd) return;
  if (fd->thread.joinable()) fd->thread.join();
  (void)hipSetDevice(fd->device);
  for (hipEvent_t ev : fd->done) if (ev) (void)hipEventDestroy(ev);
  if (fd->lender) {
    if (fd->s_copy) (void)hipStreamSynchronize(fd->s_copy);
    fd->lender->staging_busy.store(false);
  } else {
    for (int b = 0; b < 2; b++) {
      if (fd->staging_free[b]) (void)hipEventDestroy(fd->staging_free[b]);
      if (fd->staging[b]) (void)hipHostFree(fd->staging[b]);
    }
    if (fd->s_copy) (void)hipStreamDestroy(fd->s_copy);
  }
  delete fd;
}

// corpus of `segments` (host floats, in device order) with local frame offsets; async: returns with the feeder running
int corpus_from_segments(sr_model* m, std::vector<Segment> segments, const uint64_t* frame_off, uint32_t n_utts, bool async,
                         sr_corpus** out) {
  *out = nullptr;
  if (!m) return fail(SR_EINVAL, "null model handle");
  hipError_t e = hipSetDevice(m->device);
  if (e != hipSuccess) return fail(SR_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
  if (!frame_off) return fail(SR_EINVAL, "frame_off is null");
  if (frame_off[0] != 0) return fail(SR_EINVAL, "frame_off[0] must be 0");
  for (uint32_t u = 0; u < n_utts; u++) {
    if (frame_off[u + 1] < frame_off[u]) return fail(SR_EINVAL, "frame_off must be non-decreasing (utterance %u)", u);
    if (frame_off[u + 1] - frame_off[u] > 65535)
      return fail(SR_ELIMIT, "utterance %u has %llu frames; back pointers are 16 bit like the reference's Book::bkp (max 65535)",
                  u, (unsigned long long)(frame_off[u + 1] - frame_off[u]));
  }
  const uint64_t F = frame_off[n_utts], total = F * m->dim;
  uint64_t have = 0;
  for (const Segment& sg : segments) {
    if (sg.n_floats && !sg.src) return fail(SR_EINVAL, "feats is null");
    have += sg.n_floats;
  }
  if (have != total) return fail(SR_EINVAL, "feature segments hold %llu floats, the offsets ask for %llu", (unsigned long long)have, (unsigned long long)total);
  sr_corpus* c = new sr_corpus();
  std::unique_ptr<sr_corpus, int (*)(sr_corpus*)> own(c, sr_corpus_destroy);
  c->model = m; c->n_utts = n_utts; c->n_frames = F;
  srhost::corpus_register(c);
  c->frame_off.assign(frame_off, frame_off + n_utts + 1);
  srhost::corpus_adopt_spare(c);
  if ((e = c->feats.ensure((size_t)total + 64)) != hipSuccess || (e = c->d_frame_off.upload(frame_off, n_utts + 1)) != hipSuccess)
    return fail(SR_EHIP, "corpus upload: %s", hipGetErrorString(e));
  if (total == 0) { *out = own.release(); return SR_OK; }
  sr_feeder* fd = new sr_feeder();
  c->feeder = fd;  // (freed with the corpus from here on)
  fd->device = m->device; fd->dst = c->feats.p; fd->segments = std::move(segments); fd->total_floats = total;
  fd->piece_floats = kPieceBytes / sizeof(float);
  fd->n_pieces = (uint32_t)((total + fd->piece_floats - 1) / fd->piece_floats);
  fd->done.assign(fd->n_pieces, nullptr);
  // the model keeps one set of feeder resources (copy stream, two pinned 2 MiB buffers); a second upload that overlaps the
  // first gets its own
  bool expected = false;
  if (m->staging_busy.compare_exchange_strong(expected, true)) {
    fd->lender = m;
    if (!m->s_copy && (e = hipStreamCreateWithFlags(&m->s_copy, hipStreamNonBlocking)) != hipSuccess)
      return fail(SR_EHIP, "hipStreamCreate (feeder): %s", hipGetErrorString(e));
    for (int b = 0; b < 2; b++)
      if ((!m->staging[b] && (e = hipHostMalloc(reinterpret_cast<void**>(&m->staging[b]), kPieceBytes, hipHostMallocDefault)) != hipSuccess) ||
          (!m->staging_free[b] && (e = hipEventCreateWithFlags(&m->staging_free[b], hipEventDisableTiming)) != hipSuccess))
        return fail(SR_EHIP, "pinned staging buffer: %s", hipGetErrorString(e));
    fd->s_copy = m->s_copy;
    for (int b = 0; b < 2; b++) { fd->staging[b] = m->staging[b]; fd->staging_free[b] = m->staging_free[b]; }
  } else {
    if ((e = hipStreamCreateWithFlags(&fd->s_copy, hipStreamNonBlocking)) != hipSuccess)
      return fail(SR_EHIP, "hipStreamCreate (feeder): %s", hipGetErrorString(e));
    for (int b = 0; b < 2; b++)
      if ((e = hipHostMalloc(reinterpret_cast<void**>(&fd->staging[b]), std::min<uint64_t>(kPieceBytes, total * sizeof(float)), hipHostMallocDefault)) != hipSuccess ||
          (e = hipEventCreateWithFlags(&fd->staging_free[b], hipEventDisableTiming)) != hipSuccess)
        return fail(SR_EHIP, "pinned staging buffer: %s", hipGetErrorString(e));
  }
  for (uint32_t i = 0; i < fd->n_pieces; i++)
    if ((e = hipEventCreateWithFlags(&fd->done[i], hipEventDisableTiming)) != hipSuccess)
      return fail(SR_EHIP, "hipEventCreate (feeder): %s", hipGetErrorString(e));
  fd->thread = std::thread(feeder_main, fd);
  if (!async) {
    const int rc = sr_corpus_wait(c);
    if (rc != SR_OK) return rc;
  }
  *out = own.release();
  return SR_OK;
}

}  // namespace

namespace srhost {

bool corpus_upload_in_flight(const sr_corpus* c) {
  if (!c || !c->feeder) return false;
  std::lock_guard<std::mutex> lk(c->feeder->mu);
  return !c->feeder->finished;
}

int corpus_ready(sr_corpus* c, uint64_t f0, uint64_t f1, hipStream_t stream) {
  sr_feeder* fd = c ? c->feeder : nullptr;
  if (!fd || f1 <= f0) return SR_OK;
  const uint64_t D = c->model->dim;
  const uint32_t p0 = (uint32_t)(f0 * D / fd->piece_floats);
  const uint32_t p1 = (uint32_t)std::min<uint64_t>(fd->n_pieces, (f1 * D + fd->piece_floats - 1) / fd->piece_floats);
  bool landed;
  {
    std::unique_lock<std::mutex> lk(fd->mu);
    fd->cv.wait(lk, [&] { return fd->n_issued >= p1 || fd->rc != SR_OK; });
    if (fd->rc != SR_OK) return fail(fd->rc, "%s", fd->err.c_str());
    landed = fd->finished;
  }
  if (landed) {
    // Everything has landed (the feeder synchronised its stream): hand the model's copy stream, pinned buffers and the piece
    // events back NOW, not at sr_corpus_destroy -- a corpus that stays resident (a training iteration, a bench loop) would
    // otherwise keep the lease and every later sr_recognize_batch on the model would allocate its own.
    feeder_join(c);
    return SR_OK;
  }
  // copies on s_copy complete in order: the last piece of the range covers the earlier ones
  if (p1 > p0) {
    hipError_t e = hipStreamWaitEvent(stream, fd->done[p1 - 1], 0);
    if (e != hipSuccess) return fail(SR_EHIP, "hipStreamWaitEvent (feeder): %s", hipGetErrorString(e));
  }
  return SR_OK;
}

void feeder_join(sr_corpus* c) {
  if (!c || !c->feeder) return;
  feeder_free(c->feeder);
  c->feeder = nullptr;
}

}  // namespace srhost

extern "C" {

int sr_corpus_upload_async(sr_model* m, const float* feats, const uint64_t* frame_off, uint32_t n_utts, sr_corpus** out) {
  return guarded(__func__, [&]() -> int {
    if (!out) return fail(SR_EINVAL, "out is null");
    *out = nullptr;
    if (!m) return fail(SR_EINVAL, "null model handle");
    if (!frame_off) return fail(SR_EINVAL, "frame_off is null");
    std::vector<Segment> segs(1, Segment{feats, frame_off[n_utts] * m->dim});
    return corpus_from_segments(m, std::move(segs), frame_off, n_utts, true, out);
  });
}

int sr_corpus_wait(sr_corpus* c) {
  return guarded(__func__, [&]() -> int {
    if (!c) return fail(SR_EINVAL, "null corpus handle");
    sr_feeder* fd = c->feeder;
    if (!fd) return SR_OK;
    {
      std::unique_lock<std::mutex> lk(fd->mu);
      fd->cv.wait(lk, [&] { return fd->finished; });
      if (fd->rc != SR_OK) return fail(fd->rc, "%s", fd->err.c_str());  // (the feeder stays: later calls report the same error)
    }
    srhost::feeder_join(c);  // joins the thread, destroys the piece events and returns the model's staging lease
    return SR_OK;
  });
}

int sr_shard_utterances(const uint64_t* frame_off, uint32_t n_utts, uint32_t n_shards, uint32_t* shard_of_utt, uint64_t* shard_frames) {
  return guarded(__func__, [&]() -> int {
    if (!frame_off || (!shard_of_utt && n_utts)) return fail(SR_EINVAL, "null argument");
    if (n_shards == 0) return fail(SR_EINVAL, "n_shards must be positive");
    // greedy longest-processing-time: utterances by decreasing length (stable), each to the lightest shard so far
    // (lowest index on ties) -- the same deal as speechrecognition_amd/sharding.py::shard_utterances
    std::vector<uint32_t> order(n_utts);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
      return frame_off[a + 1] - frame_off[a] > frame_off[b + 1] - frame_off[b];
    });
    std::vector<uint64_t> load(n_shards, 0);
    for (uint32_t u : order) {
      const uint32_t r = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
      shard_of_utt[u] = r;
      load[r] += frame_off[u + 1] - frame_off[u];
    }
    if (shard_frames) std::copy(load.begin(), load.end(), shard_frames);
    return SR_OK;
  });
}

int sr_recognize_batch_multi(sr_model* const* models, sr_lexicon* const* lexica, uint32_t n_devices, const sr_search_params* p,
                             const float* feats, const uint64_t* frame_off, uint32_t n_utts, uint32_t* out_words,
                             uint64_t* out_word_off, uint64_t* shard_frames) {
  return guarded(__func__, [&]() -> int {
    if (!models || !lexica || n_devices == 0 || !p || !frame_off || !out_word_off) return fail(SR_EINVAL, "null argument");
    for (uint32_t d = 0; d < n_devices; d++) {
      if (!models[d] || !lexica[d] || lexica[d]->model != models[d]) return fail(SR_EINVAL, "device slot %u: lexicon does not belong to its model", d);
      if (models[d]->dim != models[0]->dim || models[d]->n_states != models[0]->n_states) return fail(SR_EINVAL, "device slot %u: model replica differs", d);
      for (uint32_t e = 0; e < d; e++)
        if (models[e] == models[d]) return fail(SR_EINVAL, "device slots %u and %u share one handle (a handle serves one host thread)", e, d);
    }
    const uint64_t F = frame_off[n_utts];
    if (F && (!feats || !out_words)) return fail(SR_EINVAL, "null buffer");
    const uint32_t D = models[0]->dim;
    std::vector<uint32_t> shard_of(n_utts);
    std::vector<uint64_t> load(n_devices);
    int rc = sr_shard_utterances(frame_off, n_utts, n_devices, shard_of.data(), load.data());
    if (rc != SR_OK) return rc;
    if (shard_frames) std::copy(load.begin(), load.end(), shard_frames);
    struct Shard {
      std::vector<uint32_t> utts;
      std::vector<uint64_t> off, woff;
      std::vector<uint32_t> words;
      int rc = SR_OK;
      std::string err;
    };
    std::vector<Shard> shards(n_devices);
    for (uint32_t u = 0; u < n_utts; u++) shards[shard_of[u]].utts.push_back(u);  // ascending inside a shard
    auto work = [&](uint32_t d) {
      Shard& sh = shards[d];
      try {
        std::vector<Segment> segs;
        sh.off.assign(1, 0);
        for (uint32_t u : sh.utts) {
          const uint64_t n = frame_off[u + 1] - frame_off[u];
          // consecutive utterances of the caller's buffer merge into one segment
          if (!segs.empty() && segs.back().src + segs.back().n_floats == feats + frame_off[u] * D) segs.back().n_floats += n * D;
          else segs.push_back(Segment{feats + frame_off[u] * D, n * D});
          sh.off.push_back(sh.off.back() + n);
        }
        sh.words.assign(std::max<uint64_t>(sh.off.back(), 1), 0);
        sh.woff.assign(sh.utts.size() + 1, 0);
        sr_corpus* c = nullptr;
        sh.rc = corpus_from_segments(models[d], std::move(segs), sh.off.data(), (uint32_t)sh.utts.size(), true, &c);
        if (sh.rc == SR_OK) {
          std::unique_ptr<sr_corpus, int (*)(sr_corpus*)> own(c, sr_corpus_destroy);
          sh.rc = sr_recognize_corpus(models[d], c, lexica[d], p, sh.words.data(), sh.woff.data(), nullptr, nullptr, nullptr);
        }
        if (sh.rc != SR_OK) sh.err = sr_last_error();  // (thread-local: carry it over to the caller's thread)
      } catch (const std::exception& e) {
        sh.rc = SR_EINTERNAL; sh.err = e.what();
      } catch (...) {
        sh.rc = SR_EINTERNAL; sh.err = "unexpected exception in a device thread";
      }
    };
    {
      srhost::ThreadGroup pool(n_devices - 1);  // joins in its destructor; a refused thread's device is driven from this thread
      for (uint32_t d = 1; d < n_devices; d++) pool.run([&work, d]() { work(d); });
      pool.run_here([&work]() { work(0); });  // the caller's thread drives the first device
      pool.wait();
    }
    for (uint32_t d = 0; d < n_devices; d++)
      if (shards[d].rc != SR_OK) return fail(shards[d].rc, "device slot %u (device %d): %s", d, models[d]->device, shards[d].err.c_str());
    // gather in corpus order
    std::vector<uint32_t> pos(n_devices, 0);
    uint64_t w = 0;
    out_word_off[0] = 0;
    for (uint32_t u = 0; u < n_utts; u++) {
      Shard& sh = shards[shard_of[u]];
      const uint32_t i = pos[shard_of[u]]++;
      for (uint64_t k = sh.woff[i]; k < sh.woff[i + 1]; k++) out_words[w++] = sh.words[k];
      out_word_off[u + 1] = w;
    }
    return SR_OK;
  });
}

}  // extern "C"
