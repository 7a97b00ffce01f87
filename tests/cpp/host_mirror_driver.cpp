// host_mirror_driver.cpp -- drives include/sr_sietill.hpp (the C++ host mirror of the reference interface)
// for tests/test_host_mirror.py.  Modes:
//   edit <file>                       lines "r1 r2 .. | h1 h2 .." -> "total sub ins del" per line (CPU only)
//   lexicon                           prints the flattened sietill digit lexicon (CPU only)
//   run <mixset> <dim> <case.bin>     recognise + align a small corpus on the GPU, print results
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "sr_sietill.hpp"

template <typename T>
static T rd(std::istream& in) {
  T v;
  in.read(reinterpret_cast<char*>(&v), sizeof v);
  return v;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "edit")) {
    std::ifstream in(argv[2]);
    std::string line;
    while (std::getline(in, line)) {
      std::vector<sr::WordIdx> ref, hyp;
      std::istringstream ss(line);
      std::string tok;
      bool right = false;
      while (ss >> tok) {
        if (tok == "|") { right = true; continue; }
        (right ? hyp : ref).push_back(std::stoul(tok));
      }
      sr::EDAccumulator ed = sr::Recognizer::editDistance(ref.data(), ref.data() + ref.size(), hyp.data(), hyp.data() + hyp.size());
      printf("%u %u %u %u\n", ed.total_count, ed.substitute_count, ed.insert_count, ed.delete_count);
    }
    return 0;
  }
  if (argc >= 2 && !strcmp(argv[1], "lexicon")) {
    sr::Lexicon lex;  // build_sietill_lexicon, sietill/Lexicon.cpp:70-85
    lex.add_word("[silence]", 1, 1, true);
    const char* names[] = {"eins", "zwei", "drei", "vier", "fuenf", "sechs", "sieben", "acht", "neun", "null", "zwo"};
    const int ns[] = {9, 9, 9, 9, 12, 9, 12, 9, 9, 9, 9};
    for (int i = 0; i < 11; i++) lex.add_word(names[i], ns[i], 2);
    printf("%zu %u %zu %zu\n", lex.num_words(), lex.num_states(), lex.silence_idx(), lex["sieben"]);
    for (size_t w = 0; w < lex.num_words(); w++) {
      auto const& a = lex.get_automaton_for_word(w);
      for (size_t i = 0; i < a.num_states(); i++) printf("%u ", a[i]);
      printf("\n");
    }
    sr::TdpModel tdp(lex.get_silence_automaton().first_state(), 3.0, 0.0, 30.0);
    printf("%g %g %g %g %g\n", tdp.score(0, 0), tdp.score(5, 0), tdp.score(5, 1), tdp.score(5, 2), tdp.score(5, 3));
    return 0;
  }
  if (argc >= 5 && !strcmp(argv[1], "features")) {  // features <file.mm2> <normalization.bin|-> <out.f32>
    std::vector<float> f = sr::read_feature_file(argv[2]);
    sr::FeaturePostProcessor pp;
    if (strcmp(argv[3], "-") && !pp.read_normalization_file(argv[3])) { printf("error normalization\n"); return 2; }
    pp.process_features(f);
    std::ofstream out(argv[4], std::ios::binary);
    out.write(reinterpret_cast<const char*>(f.data()), sizeof(float) * f.size());
    // alignment dump round trip through the reference's file format
    std::vector<sr::AlignmentItem> ali(7), back;
    for (size_t i = 0; i < ali.size(); i++) { ali[i].count = 1; ali[i].state = (uint16_t)(3 * i); ali[i].weight = 1.0f; }
    { std::ofstream d(std::string(argv[4]) + ".dump", std::ios::binary); sr::write_alignment(d, ali, 1); }
    size_t ma = 0;
    { std::ifstream d(std::string(argv[4]) + ".dump", std::ios::binary); sr::read_alignment(d, back, ma); }
    bool same = ma == 1 && back.size() == ali.size();
    for (size_t i = 0; same && i < ali.size(); i++) same = back[i].state == ali[i].state && back[i].count == 1 && back[i].weight == 1.0f;
    printf("%zu frames x %zu; dump %s, item %zu bytes\n", f.size() / pp.n_features_total(), pp.n_features_total(), same ? "ok" : "BAD",
           sizeof(sr::AlignmentItem));
    return 0;
  }
  if (argc >= 5 && !strcmp(argv[1], "run")) {
    try {
      const size_t dim = std::stoul(argv[3]);
      std::ifstream in(argv[4], std::ios::binary);
      sr::Lexicon lex;
      const uint32_t n_words = rd<uint32_t>(in);
      std::vector<std::pair<uint16_t, uint16_t>> ws(n_words);
      for (auto& w : ws) { w.first = rd<uint16_t>(in); w.second = rd<uint16_t>(in); }
      const uint32_t sil = rd<uint32_t>(in);
      for (uint32_t w = 0; w < n_words; w++) lex.add_word("w" + std::to_string(w), ws[w].first, ws[w].second, w == sil);
      const double tl = rd<double>(in), tf = rd<double>(in), ts = rd<double>(in), beam = rd<double>(in), wp = rd<double>(in);
      const int kernel = (int)rd<uint32_t>(in);
      sr::MixtureModel mm(argv[2], dim, sr::MixtureModel::NO_POOLING, true, 0, kernel);
      sr::TdpModel tdp(lex.get_silence_automaton().first_state(), tl, tf, ts);
      sr::Recognizer rec(lex, mm, tdp, beam, wp);
      sr::Corpus corpus(dim);
      const uint32_t n_utts = rd<uint32_t>(in);
      std::vector<std::vector<float>> feats(n_utts);
      for (uint32_t u = 0; u < n_utts; u++) {
        const uint32_t T = rd<uint32_t>(in), n_ref = rd<uint32_t>(in);
        std::vector<sr::WordIdx> ref(n_ref);
        for (auto& r : ref) r = rd<uint32_t>(in);
        feats[u].resize((size_t)T * dim);
        in.read(reinterpret_cast<char*>(feats[u].data()), sizeof(float) * feats[u].size());
        corpus.add_segment(feats[u].data(), T, ref);
      }
      sr::RecognitionStats st = rec.recognize(corpus);
      for (auto const& h : st.hypotheses) {
        printf("hyp");
        for (auto w : h) printf(" %zu", w);
        printf("\n");
      }
      printf("stats %u %u %u %u %zu %zu\n", st.errors.total_count, st.errors.substitute_count, st.errors.insert_count,
             st.errors.delete_count, st.ref_words, st.sentence_errors);
      // the same corpus sharded over device handles (here: three replicas on device 0, the recogniser's own + two more)
      {
        sr::RecognitionStats st3 = rec.recognize(corpus, std::vector<int>{0, 0, 0});
        bool same = st3.hypotheses == st.hypotheses && st3.errors.total_count == st.errors.total_count;
        uint64_t frames = 0;
        for (uint64_t f : st3.frames_per_device) frames += f;
        printf("multi %s %zu %llu\n", same ? "same" : "DIFFERENT", st3.frames_per_device.size(), (unsigned long long)frames);
        sr::RecognitionStats st3b = rec.recognize(corpus, std::vector<int>{0, 0, 0});  // replicas are kept and reused
        printf("multi_again %s\n", st3b.hypotheses == st.hypotheses ? "same" : "DIFFERENT");
      }
      // single-sequence entry points on utterance 0
      std::vector<sr::WordIdx> one;
      rec.recognizeSequence_pruned(feats[0].data(), feats[0].size() / dim, one);
      printf("one");
      for (auto w : one) printf(" %zu", w);
      printf("\n");
      mm.prepare_sequence(feats[0].data(), feats[0].size() / dim);
      printf("score %.17g %.17g\n", mm.score(0, 0), mm.score(feats[0].size() / dim - 1, lex.num_states() - 1));
      const uint32_t n_aut = rd<uint32_t>(in);
      sr::MarkovAutomaton aut;
      for (uint32_t i = 0; i < n_aut; i++) aut.states.push_back(rd<uint16_t>(in));
      sr::Aligner al(mm, tdp);
      std::vector<sr::AlignmentItem> ali;
      const double c_full = al.align_sequence_full(feats[0].data(), feats[0].size() / dim, aut, ali);
      printf("align %.17g", c_full);
      for (auto const& it : ali) printf(" %u", it.state);
      printf("\n");
      const double c_pr = al.align_sequence_pruned(feats[0].data(), feats[0].size() / dim, aut, ali, 30.0);
      printf("alignp %.17g", c_pr);
      for (auto const& it : ali) printf(" %u", it.state);
      printf("\n");
      // training-side callers: corpus-wide re-alignment against the transcriptions + average AM score
      sr::Trainer trainer(lex, mm, tdp, 40.0, true);
      std::vector<sr::AlignmentItem> corpus_ali;
      std::vector<double> costs;
      trainer.realign(corpus, corpus_ali, &costs);
      printf("realign");
      for (double c : costs) printf(" %.17g", c);
      printf("\nrealign_states");
      for (auto const& it : corpus_ali) printf(" %u", it.state);
      printf("\namscore %.17g\n", trainer.calc_am_score(corpus, corpus_ali));
      // bigram search mirror (Teaching::LinearSearch): the digit-style lexicon as a linear lexicon, a fixed dense LM
      {
        const uint32_t W = (uint32_t)lex.num_words();
        std::vector<std::vector<uint16_t>> linear(W);
        for (uint32_t w = 0; w < W; w++) linear[w] = lex.get_automaton_for_word(w).states;
        std::vector<float> lm((size_t)W * W);
        for (uint32_t w = 0; w < W; w++)
          for (uint32_t h = 0; h < W; h++) lm[(size_t)w * W + h] = 2.0f + 0.5f * (float)((7 * w + 3 * h) % 11);
        const float btdp[2][4] = {{3.0f, 0.0f, 30.0f, 5.0f}, {1.0f, 0.0f, 40.0f, 2.0f}};
        sr::LinearSearch search(mm, linear, sil, lm, btdp, 150.0f, 20.0f);
        std::vector<sr::LinearSearch::Traceback> res;
        search.recognize(corpus, res);
        for (auto const& tb : res) {
          printf("bigram");
          for (auto const& it : tb) printf(" %u:%.9g:%u", it.word, it.score, it.time);
          printf("\n");
        }
      }
    } catch (std::exception const& e) {
      printf("error %s\n", e.what());
      return 2;
    }
    return 0;
  }
  fprintf(stderr, "usage: see source\n");
  return 1;
}
