"""The reference-side bindings shown in INTEGRATION.md (integration/GpuMixtureScorer.hpp, integration/GpuTrainer.hpp) must
compile against the reference's own headers AND link against the reference's own objects + libsrgpu.so (every symbol the
stubs use on either side resolved).  Build container only: /root/reference does not exist on the GPU box."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/sietill"
REF_OBJ = os.path.join(ROOT, "oracle", "_ref")

TU = r"""
#include "GpuMixtureScorer.hpp"
#include "GpuTrainer.hpp"
#include "Config.hpp"

// What a maintainer's SieTill.cpp would do with the two bindings (never executed here: there is no GPU in the build container;
// linking is the test).  Every member function of both stubs is instantiated.
int drive(Configuration const& config, Corpus const& corpus, Lexicon const& lexicon, std::string const& mixture_path) {
  GpuMixtureScorer scorer(mixture_path, corpus.get_features_per_timeframe(), (int)MixtureModel::NO_POOLING, true, 0);
  TdpModel tdp_model(config, lexicon.get_silence_automaton().first_state());
  std::pair<FeatureIter, FeatureIter> seq = corpus.get_feature_sequence(0);
  scorer.prepare_sequence(seq.first, seq.second);
  double s = scorer.score(seq.first, 0);
  std::vector<std::vector<WordIdx> > recognized;
  gpu_recognize(scorer, lexicon, tdp_model, 3.0, 0.0, 30.0, 200.0, 10.0, corpus, corpus.get_corpus_size(), recognized);
  GpuMixtureScorer second(mixture_path, corpus.get_features_per_timeframe(), (int)MixtureModel::NO_POOLING, true, 1);
  std::vector<GpuMixtureScorer*> replicas;
  replicas.push_back(&scorer); replicas.push_back(&second);
  gpu_recognize(replicas, lexicon, tdp_model, 3.0, 0.0, 30.0, 200.0, 10.0, corpus, corpus.get_corpus_size(), recognized);
  GpuTrainer trainer(scorer, lexicon, tdp_model, 3.0, 0.0, 30.0, 1, 50.0, true);
  std::vector<MarkovAutomaton> automata;
  for (SegmentIdx i = 0; i < corpus.get_corpus_size(); i++) {
    std::pair<WordIter, WordIter> w = corpus.get_word_sequence(i);
    automata.push_back(trainer.build_segment_automaton(w.first, w.second));
  }
  Alignment alignment;
  std::vector<double> costs = trainer.realign(corpus, automata, alignment);
  s += trainer.calc_am_score(corpus, alignment) + costs[0];
  trainer.accumulate(corpus, alignment, false, true);
  trainer.write(mixture_path + ".next");
  sr_model* next = trainer.finalize((int)MixtureModel::NO_POOLING, true, 0);
  sr_model_destroy(next);
  return s > 0.0;
}

int main(int argc, char** argv) {
  if (argc < 100) return 0;   // (link test)
  Configuration config(argv[1]);
  Corpus corpus;
  Lexicon lexicon = build_sietill_lexicon();
  return drive(config, corpus, lexicon, argv[2]);
}
"""


def _flags():
    return ["--std=c++11", "-Wall", "-msse", "-msse2", "-msse3", "-fopenmp", "-include", "emmintrin.h", "-I" + REF,
            "-I" + os.path.join(REF, "rapidjson", "include"), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration")]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "FeatureScorer.hpp")), reason="reference sources not present")
def test_reference_side_stubs_compile_against_reference_headers(tmp_path):
    tu = tmp_path / "tu.cpp"
    tu.write_text(TU)
    r = subprocess.run(["g++", "-fsyntax-only"] + _flags() + [str(tu)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "FeatureScorer.hpp")), reason="reference sources not present")
def test_reference_side_stubs_link_against_reference_objects_and_libsrgpu(tmp_path):
    from oracle import pyoracle
    from speechrecognition_amd import build

    lib = build.build()
    pyoracle.build()   # also builds oracle/_ref (the reference's own translation units, compiled where they lie)
    objs = [o for o in sorted(glob.glob(os.path.join(REF_OBJ, "*.o"))) if not o.endswith("ref_driver.o")]
    assert any(o.endswith("Lexicon.o") for o in objs) and any(o.endswith("Corpus.o") for o in objs), objs
    tu = tmp_path / "tu.cpp"
    tu.write_text(TU)
    exe = tmp_path / "sietill_gpu_linktest"
    cmd = ["g++"] + _flags() + [str(tu)] + objs + ["-o", str(exe), "-L" + os.path.dirname(lib), "-lsrgpu",
                                                   "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # the binary starts (dynamic loader resolves libsrgpu.so and the HIP runtime) and leaves through the link-test exit
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])


# The scorer seam under the reference's own OpenMP loop (VERDICT r3 missing #4): Recognizer::recognize runs
# recognizeSequence_pruned -- i.e. scorer_.prepare_sequence + scorer_.score -- for a different segment on every thread
# through ONE scorer object (Recognizer.cpp:46,104).  The build container has no GPU, so the five libsrgpu entry points the
# binding calls are answered here by the reference's own MixtureModel (with a yield inside the "device" call to widen any
# window); what is under test is the binding's per-thread sequence state and its device mutex.
OMP_TU = r"""
#include <sched.h>
#include <omp.h>
#include <atomic>
#include <cstdio>
#include <random>
#include "GpuMixtureScorer.hpp"
#include "Config.hpp"
#include "Mixtures.hpp"
#include "Recognizer.hpp"

static MixtureModel* g_mm = NULL;
static size_t g_dim = 0, g_states = 0;
static std::atomic<int> g_inside(0), g_overlaps(0), g_calls(0);
struct sr_model { int unused; };
extern "C" {
const char* sr_last_error(void) { return "mock"; }
int sr_model_load_mixset(const char*, uint32_t, int, int, int, sr_model** out) { *out = new sr_model(); return SR_OK; }
int sr_model_destroy(sr_model* m) { delete m; return SR_OK; }
int sr_model_info(const sr_model*, uint32_t* dim, uint32_t* n_states, uint64_t* n_dens) {
  *dim = g_dim; *n_states = g_states; *n_dens = 0; return SR_OK;
}
int sr_score_frames(sr_model*, const float* feats, uint64_t n_frames, int, double* out) {
  if (g_inside.fetch_add(1) != 0) g_overlaps++;   // srgpu.h: one host thread at a time per handle
  g_calls++;
  FeatureIter it(const_cast<float*>(feats), g_dim);
  for (uint64_t t = 0; t < n_frames; t++, ++it) {
    for (size_t s = 0; s < g_states; s++) out[t * g_states + s] = g_mm->score(it, s);
    if ((t & 7) == 0) sched_yield();
  }
  g_inside.fetch_sub(1);
  return SR_OK;
}
}

int main(int argc, char** argv) {
  Configuration config((std::string(argv[1])));
  const size_t dim = 13, n_seq = 48;
  Lexicon lexicon;
  lexicon.add_word("[silence]", 1, 1, true);
  for (int w = 0; w < 6; w++) lexicon.add_word("w" + std::to_string(w), 3, 1, false);
  MixtureModel mm(config, dim, lexicon.num_states(), MixtureModel::NO_POOLING, true);
  g_mm = &mm; g_dim = dim; g_states = lexicon.num_states();
  TdpModel tdp(config, lexicon.get_silence_automaton().first_state());
  std::mt19937 rng(5);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<std::vector<float> > seqs(n_seq);
  for (size_t s = 0; s < n_seq; s++) {
    const size_t T = 20 + (rng() % 60);
    seqs[s].assign(T * dim + 4, 0.0f);
    for (size_t i = 0; i < T * dim; i++) seqs[s][i] = (float)(nd(rng) * 1.5);
  }
  // serial, unmodified: Recognizer over the reference's MixtureModel
  std::vector<std::vector<WordIdx> > want(n_seq), got(n_seq);
  {
    Recognizer rec(config, lexicon, mm, tdp);
    for (size_t s = 0; s < n_seq; s++) {
      const size_t T = (seqs[s].size() - 4) / dim;
      rec.recognizeSequence_pruned(FeatureIter(seqs[s].data(), dim), FeatureIter(seqs[s].data() + T * dim, dim), want[s]);
    }
  }
  // the reference's loop shape (Recognizer.cpp:46-56) over ONE GpuMixtureScorer, 8 threads
  GpuMixtureScorer scorer("unused", dim, (int)MixtureModel::NO_POOLING, true, 0);
  Recognizer rec(config, lexicon, scorer, tdp);
  omp_set_num_threads(8);
  for (int round = 0; round < 3; round++) {
#pragma omp parallel for ordered schedule(dynamic)
    for (size_t s = 0; s < n_seq; s++) {
      const size_t T = (seqs[s].size() - 4) / dim;
      std::vector<WordIdx> words;
      rec.recognizeSequence_pruned(FeatureIter(seqs[s].data(), dim), FeatureIter(seqs[s].data() + T * dim, dim), words);
      got[s] = words;
    }
    size_t bad = 0, n_words = 0;
    for (size_t s = 0; s < n_seq; s++) { bad += got[s] != want[s]; n_words += want[s].size(); }
    std::printf("round %d: %zu of %zu sequences differ, %zu words, device calls %d, overlapping device calls %d\n", round, bad, n_seq,
                n_words, g_calls.load(), g_overlaps.load());
    if (bad || g_overlaps.load()) return 1;
  }
  return 0;
}
"""


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "FeatureScorer.hpp")), reason="reference sources not present")
def test_scorer_seam_is_safe_under_the_reference_openmp_loop(tmp_path):
    import json
    import numpy as np
    from oracle import pyoracle
    from speechrecognition_amd import synth

    pyoracle.build()
    objs = [o for o in sorted(glob.glob(os.path.join(REF_OBJ, "*.o"))) if not o.endswith("ref_driver.o")]
    lex = synth.make_lexicon(6, 3, 1)
    spec = synth.make_mixset(lex.n_states, 3, 13, seed=3)
    mix = tmp_path / "m.mix"
    synth.write_mixset(str(mix), spec)
    cfg = tmp_path / "c.json"
    cfg.write_text(json.dumps({"action": "recognize", "verbosity": "noLog", "load-mixtures-from": str(mix), "am-threshold": 60.0,
                               "word-penalty": 10.0, "tdp-loop": 3.0, "tdp-forward": 0.0, "tdp-skip": 30.0}))
    tu = tmp_path / "omp_tu.cpp"
    tu.write_text(OMP_TU)
    exe = tmp_path / "omp_seam"
    r = subprocess.run(["g++", "-O1"] + _flags() + [str(tu)] + objs + ["-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([str(exe), str(cfg)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "round 2: 0 of 48 sequences differ" in r.stdout and "overlapping device calls 0" in r.stdout
    assert ", device calls 144, overlapping" in r.stdout  # 3 rounds x 48 sequences, one prepare_sequence each
