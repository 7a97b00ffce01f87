// traceback.h -- the end of Recognizer::recognizeSequence_pruned (sietill/Recognizer.cpp:222-231), shared by the three
// search kernels, the device re-walk (sr_traceback_corpus) and the host walk (sr_traceback_words):
//
//     t = T; while (t > 0) { if (traceback[t].word != silence) push(word); t = traceback[t].bkp; }  reverse
//
// In a traceback the reference (or one of the kernels here) wrote, bkp is the frame before the word started
// (`bkp = t - 1` of an earlier frame, Recognizer.cpp:140; 16-bit truncation keeps it below t), so t falls strictly and at
// most T words are pushed.  The walk does not take that on trust: an entry whose bkp is not below its frame, whose word is
// not a word of the lexicon, or a word list that outgrows its buffer ends the walk with kTbCorrupt, and the caller raises
// out_flags bit 2 -> SR_ECORRUPT.  (Round 2: a timing-probe build with the word-end reduction stubbed out left
// traceback entries unwritten; the unguarded walk followed the stale bytes, never reached t = 0 and wrote words past the
// end of out_words until it left the allocation -- the "Memory access fault" of gpurun_out/ab_fp2.log)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srgpu {

static constexpr uint32_t kFlagSlowPath = 1u;   // out_flags: the sequential boundary replay was taken (informational)
static constexpr uint32_t kFlagReplay = 2u;     // the fast kernel met a negative emission cost: decode_kernel<REPLAY> redoes the utterance
static constexpr uint32_t kFlagCorrupt = 4u;    // the traceback did not walk: no words are reported, the entry point returns SR_ECORRUPT
static constexpr uint32_t kTbCorrupt = 0xFFFFFFFFu;

// words[0 .. return) = the recognised words in time order; load_word(t) / load_bkp(t) read traceback[t], t = 1..T.
template <class LoadWord, class LoadBkp>
__host__ __device__ inline uint32_t walk_traceback(uint32_t T, uint32_t silence_word, uint32_t n_words, LoadWord load_word,
                                                   LoadBkp load_bkp, uint32_t* words, uint32_t capacity) {
  uint32_t n = 0, t = T;
  while (t > 0) {
    const uint32_t w = load_word(t);
    if (w >= n_words) return kTbCorrupt;
    if (w != silence_word) {
      if (n >= capacity) return kTbCorrupt;
      words[n++] = w;
    }
    const uint32_t b = load_bkp(t);
    if (b >= t) return kTbCorrupt;
    t = b;
  }
  for (uint32_t i = 0; i < n / 2; i++) { const uint32_t x = words[i]; words[i] = words[n - 1 - i]; words[n - 1 - i] = x; }
  return n;
}

}  // namespace srgpu
