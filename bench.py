#!/usr/bin/env python3
"""bench.py -- frames/s scored + decoded on MI355X for BASELINE.json's headline configuration.

Workload (BASELINE.json configs[2]): 4000 tied states (silence + 1333 three-state words),
32-mixture diagonal GMM (128 000 densities), 39-dim float32 features, a batch of 1000 synthetic
utterances (lengths U{200..400}, i.i.d. N(0,1) frames) per GPU, beam (am-threshold) 200, word penalty 10,
TDP 3/0/30.  A "step" is one full pass over the resident batch: exact GMM scoring of every (frame, state)
(default: fp16 MFMA prefilter + FP64 refinement, bit-identical to MixtureModel::score) + beam Viterbi decode
of every utterance + recognised words back on the host.  Features are resident in HBM before the timed region
(sr_corpus_upload); results leave the device inside it.

Multi-GPU (`--gpus N`): ONE batch of N x 1000 utterances (or `--total-utts T`: configs[3] = `--gpus 8 --total-utts 10000`)
is generated from a shared seed and dealt to the ranks by frames (greedy longest-processing-time, the same deal
sr_shard_utterances makes); one process per GPU (torchrun), each rank holds a full model replica and decodes its
shard; there is no data-path collective (utterances are independent, Recognizer.cpp:46-47).  torch.distributed
is used only for the barrier and the max-over-ranks of the step time.  Started without torchrun, `--gpus N`
starts it as a child process (before anything touches the GPU) and relays its JSON line.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (GMM kernel, timed with
HIP events on its launch stream by libsrgpu's sr_profile_*) and, at N=1, `cpu_baseline` (the CPU oracle's
Recognizer::recognize loop on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD's MI355X FP64 matrix figure; the local guide has no FP64 row
FRAME_SHIFT_S = 0.010         # 10 ms frames (sietill/SignalAnalysis.cpp:49, Corpus.cpp:92)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=1000, help="utterances per GPU (the batch holds utts x gpus)")
    ap.add_argument("--total-utts", type=int, default=0, help="size of the whole batch instead (strong scaling: fixed as N grows)")
    ap.add_argument("--words", type=int, default=1333, help="three-state words (states = 1 + 3*words)")
    ap.add_argument("--extra-states-last", type=int, default=0, help="extra states of the last word (configs[4]: --words 2666 "
                    "--extra-states-last 1 --mix 64 gives 8000 states)")
    ap.add_argument("--config", choices=["cfg3", "cfg4", "cfg5"], default=None,
                    help="presets: cfg3 = BASELINE configs[2] (default), cfg4 = configs[3] (--total-utts 10000), "
                         "cfg5 = configs[4] (8000 states x 64, bigram search)")
    ap.add_argument("--mix", type=int, default=32)
    ap.add_argument("--beam", type=float, default=200.0)
    ap.add_argument("--sum-mode", action="store_true",
                    help="score with max-approx false: -log sum_d exp(-score_d) per state (Mixtures.cpp:719-728) instead of the minimum; "
                         "the library's default kernel for such a model is the dense FP64-MFMA one (SR_GMM_DEFAULT)")
    ap.add_argument("--kernel", choices=["mfma", "exact", "prefilter", "default"], default="prefilter",
                    help="GMM scoring path: prefilter (fp16 MFMA candidate pass + exact FP64 refinement, bit-exact scores), "
                         "mfma (dense FP64 MFMA, ~1e-15), exact (dense FP64 VALU, bit-exact)")
    ap.add_argument("--decoder", choices=["zerogram", "bigram"], default="zerogram",
                    help="zerogram: Recognizer::recognizeSequence_pruned (the headline config); bigram: Teaching::LinearSearch "
                         "with a seeded dense bigram table (BASELINE configs[4]: use --words 2666 --mix 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-mfma", action="store_true", help="skip the one untimed step through the dense FP64 MFMA kernel")
    ap.add_argument("--no-boundary", action="store_true", help="skip the untimed host-buffers-in / words-out steps (sr_recognize_batch)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals "
                    "where several ranks share one GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall budget of the all-core cpu_baseline leg (the one-thread leg adds one short utterance)")
    args = ap.parse_args()
    if args.config == "cfg4":
        args.total_utts = 10000
    if args.sum_mode and args.kernel == "prefilter":
        args.kernel = "default"
    if args.config == "cfg5":
        args.words, args.extra_states_last, args.mix, args.decoder = 2666, 1, 64, "bigram"
    return args


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without torchrun: start it as a CHILD process (never exec: nothing here has touched the GPU
    yet, and it must stay that way) and relay its output and exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and os.environ.get("SR_BENCH_FORCE_DIST") != "1":
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}\n")
        sys.exit(2)
    # Only the JSON line may reach stdout: libraries write banners there (RCCL prints its version block when the first
    # communicator comes up), so fd 1 points at stderr for the whole run and the result goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("SR_BENCH_FORCE_DIST") == "1"  # (the env switch rehearses the RCCL path with one rank)
    n_dev = torch.cuda.device_count()
    device = local_rank if args.dist_backend == "nccl" else local_rank % max(1, n_dev)  # gloo rehearsal may share a GPU
    torch.cuda.set_device(device)
    if distributed:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=args.dist_backend)

    from speechrecognition_amd import capi, sharding, synth

    D = 39
    lex = synth.make_lexicon(args.words, 3, 1, extra_states_last=args.extra_states_last)
    S = lex.n_states
    tdp, wp = (3.0, 0.0, 30.0), 10.0
    tmp = tempfile.mkdtemp(prefix=f"srbench{rank}_")
    mixset_path = os.path.join(tmp, "model.mix")
    spec = synth.make_mixset(S, args.mix, D, seed=23)       # same model on every rank (replicated, read-only)
    synth.write_mixset(mixset_path, spec)
    # ONE batch for the whole job, the same on every rank (shared seed), dealt to the ranks by frames (greedy LPT)
    total_utts = args.total_utts or args.utts * world
    all_feats, all_off = synth.make_batch(total_utts, 200, 400, D, seed=7)
    shards = sharding.shard_utterances(all_off, world)
    shard_frames = [int(np.diff(all_off.astype(np.int64))[sh].sum()) for sh in shards]
    feats, frame_off = (all_feats, all_off) if world == 1 else sharding.take_shard(all_feats, all_off, shards[rank])
    del all_feats
    n_frames = int(frame_off[-1])
    word_off, automaton, sil_state = lex.flatten()
    kernel = {"mfma": capi.GMM_MFMA, "exact": capi.GMM_EXACT, "prefilter": capi.GMM_PREFILTER, "default": capi.GMM_DEFAULT}[args.kernel]
    if args.kernel == "default":  # what SR_GMM_DEFAULT resolves to for this model (srgpu.h): the report names the kernel that ran
        args.kernel = "mfma" if args.sum_mode else "prefilter"
    elif args.sum_mode and args.kernel == "prefilter":
        args.kernel = "exact"     # SR_GMM_PREFILTER on a sum-mode model is scored by the exact kernel (same bits as SR_GMM_EXACT)

    model = capi.Model.from_mixset(mixset_path, D, capi.POOL_NONE, not args.sum_mode, device=device)
    lexh = model.lexicon(word_off, automaton, lex.silence_idx, tdp, sil_state)
    corpus = model.upload(feats, frame_off)  # inputs resident in HBM before timing starts

    bg = None
    if args.decoder == "bigram":
        # dense bigram table: rows of p(. | h) from a symmetric Dirichlet(1), scale 1; transition scores of
        # the reference's example set-up (example-setup/config/recognition-triphones-lda-pruned.config:46-57)
        lm_rng = np.random.default_rng(99)
        nW = lex.n_words
        lm = np.empty((nW, nW), np.float32)
        for h0 in range(0, nW, 256):
            p = lm_rng.dirichlet(np.ones(nW), size=min(256, nW - h0))
            lm[:, h0:h0 + p.shape[0]] = (-np.log(np.maximum(p, 1e-30))).T
        bg_tdp = np.array([[3.0, 0.0, 3.0, 150.0], [0.0001, 3.0, np.inf, 15.0]], np.float32)
        bg = model.bigram(word_off, automaton, lex.silence_idx, lm, bg_tdp)

    def step():
        if bg is not None:
            w, _, _, off = corpus.recognize_bigram(bg, args.beam, capi.FLT_MAX, kernel)
            return w, off
        return corpus.recognize(lexh, args.beam, wp, kernel)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        words, woff = step()
    model.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        words, woff = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = model.profile_read()
    model.profile(False)
    # north_star's "MFMA utilisation on the GMM step": the headline path no longer executes the dense FP64 contraction, so the
    # dense FP64-MFMA kernel (same scores to 1e-9) is timed once beside it, outside the timed region, for that figure
    dense_mfma = None
    if rank == 0 and world == 1 and args.kernel == "prefilter" and bg is None and not args.no_dense_mfma and not args.sum_mode:
        corpus.recognize(lexh, args.beam, wp, capi.GMM_MFMA)  # builds the packing
        model.profile(True)
        torch.cuda.synchronize()
        corpus.recognize(lexh, args.beam, wp, capi.GMM_MFMA)
        torch.cuda.synchronize()
        pm = model.profile_read()
        model.profile(False)
        if pm["gmm_ms"] > 0:
            tf = pm["gmm_flops"] / (pm["gmm_ms"] * 1e-3) / 1e12
            dense_mfma = {"kernel": "gmm_mfma_kernel (v_mfma_f64_16x16x4_f64), one step, not part of `value`", "ms": pm["gmm_ms"],
                          "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                          "flops": pm["gmm_flops"]}

    # The drop-in boundary hands over HOST buffers (the reference's own timed region starts from host memory, Recognizer.cpp:45): a few
    # untimed steps through sr_recognize_batch -- features over PCIe by the asynchronous feeder, words back -- beside the resident
    # rate.  Never `value`.
    boundary = None
    if rank == 0 and world == 1 and bg is None and not args.no_boundary:
        model.recognize_batch(lexh, feats, frame_off, args.beam, wp, kernel)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        nb = 3
        for _ in range(nb):
            wb, ob = model.recognize_batch(lexh, feats, frame_off, args.beam, wp, kernel)
        dt = (time.perf_counter() - tb) / nb
        boundary = {"ms_per_step_host_in_words_out": dt * 1e3, "frames_per_s": n_frames / dt, "steps": nb,
                    "what": "sr_recognize_batch: host float32 features in (asynchronous feeder: 2 MiB pinned pieces on a copy stream, scoring "
                            "starts on the first 1/24), words out; includes corpus set-up and tear-down per step.  Reported beside `value`, never as it",
                    "words_equal_resident": bool(np.array_equal(wb, words) and np.array_equal(ob, woff))}

    elapsed, total_frames = sharding.reduce_timing(elapsed, n_frames, dist if distributed else None,
                                                   torch.device("cuda", device) if args.dist_backend == "nccl" else None)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_frames * args.steps / elapsed
        gmm_s = prof["gmm_ms"] * 1e-3
        achieved = prof["gmm_flops"] / gmm_s / 1e12 if gmm_s > 0 else 0.0
        out = {
            "metric": "frames/sec scored+decoded",
            "value": value,
            "unit": "frames/s",
            "xRT": value * FRAME_SHIFT_S,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if args.total_utts else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{S} states x {args.mix}-mix diag GMM ({S * args.mix} densities), 39-d float32 frames, "
                            f"one batch of {total_utts} utterances U{{200..400}} frames ({int(all_off[-1])} frames), beam {args.beam:g}, "
                            f"exact scoring of every (frame, state) + beam Viterbi",
                "states": S, "mixtures": args.mix, "feat_dim": D, "utterances_total": total_utts,
                "utterances_rank0": len(frame_off) - 1, "frames_rank0": n_frames, "words": lex.n_words,
                "trellis_positions": int(word_off[-1]), "gmm_kernel": args.kernel,
                "mixture_score": "sum: -log sum exp(-score_d) (max-approx false, Mixtures.cpp:719-728; device exp/log: 1e-12 relative)" if args.sum_mode
                                 else "max-approx: min over the densities (Mixtures.cpp:696-713), bit-exact",
                "parallelism": f"utterance-shard x{world} (greedy LPT by frames), no collective",
                "shard_imbalance": max(shard_frames) / (sum(shard_frames) / world),  # heaviest shard / mean: what strong scaling can lose
            },
            "roofline": gmm_roofline(args, prof, n_frames, D, S),
            "search": search_report(args, prof, S, int(word_off[-1]), padded_slots(word_off, automaton, lex.silence_idx, sil_state), n_frames, len(frame_off) - 1, lexh.describe()),
            "recognised_words_rank0": int(woff[-1]),
            "build": build_identity(),
        }
        # what of the search is NOT hidden behind scoring: with one score chunk per step the search follows the scoring (exposed = all of
        # it); with several chunks it overlaps the next chunk's scoring on a second stream, and step - GMM is what shows in the step
        out["search"]["exposed_ms"] = ms_per_step - prof["gmm_ms"] / max(1, args.steps)
        if boundary:
            out["boundary"] = boundary
        if args.kernel == "prefilter":
            out.update(prefilter_report(args, prof, n_frames, D, S))
        if dense_mfma:
            out["gmm_dense_fp64_mfma"] = dense_mfma
        if bg is not None:
            out["config"]["workload"] = out["config"]["workload"].replace("beam Viterbi", "bigram linear-lexicon beam search "
                                                                          "(Teaching::LinearSearch, parity unpinned)")
            out["search"]["kernel"] = "bigram_kernel (short-word lexica: state hypotheses in registers, one lane per word and its silence copy; viterbi_bigram.hip)"
            out["search"]["bound"] = ("instruction issue: 11 barriers per frame, ~1 600 vector instructions per wave between them and ~135 spilled scalar "
                                      "registers reloaded through v_readlane; the oldest wave of a SIMD runs a step in a third of the time and waits for "
                                      "the other three at the barrier (profiles/r4_bigram_steps.txt, r4_pmc_cfg5.txt); not memory")
            out["search"].pop("network", None)
        if world == 1 and not args.no_cpu_baseline:
            if bg is not None:
                out["cpu_baseline"] = cpu_baseline_bigram(args, mixset_path, lex, lm, bg_tdp, feats, frame_off, words, woff)
            else:
                out["cpu_baseline"] = cpu_baseline(args, mixset_path, lex, tdp, wp, feats, frame_off, words, woff)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    corpus.close()
    if bg is not None:
        bg.close()
    lexh.close()
    model.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


FP64_VALU_UNFUSED_PEAK = 39.3  # Tflop/s: one v_add_f64 / v_mul_f64 per lane and cycle (profiles/r1_fp64_rate_microbench.txt: 36.8)
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA (MI355X_MICROARCH.md); ~1250-1480 sustained on random data (DVFS)


def gmm_roofline(args, prof, n_frames, D, S):
    """Roofline of the dominant kernel of the step.

    mfma / exact: the dense scoring kernel, SURVEY 8(d): 4*D*C flops per frame against the FP64 matrix peak.
    prefilter:    the FP64 refinement kernel dominates.  Its algorithmic work is ONE exact density evaluation per
                  (frame, state) -- 4*D unfused flops, the reference's operation order forbids FMA -- against the
                  FP64 vector pipe (bound "valu": the contract's enum has no name for it; the dense-FP64-equivalent
                  rate of the whole step is under "gmm_step").
    """
    launches = max(1, prof["gmm_launches"])
    steps = max(1, args.steps)
    if args.kernel == "prefilter":
        # `n_frames` is what ONE STEP scores; a corpus whose score table needs several chunks takes several launches per step,
        # so the step's flops go over the step's kernel time (= the launch-weighted mean: flops of a launch / its duration)
        ms = prof["refine_ms"] / steps
        flops = 4.0 * D * S * n_frames
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {"kernel": "gmm_refine_kernel", "bound": "valu", "achieved": achieved, "peak": FP64_VALU_UNFUSED_PEAK,
                "unit": "TFLOP/s", "frac": achieved / FP64_VALU_UNFUSED_PEAK, "traffic": pmc_traffic(args, n_frames, "gmm_refine_kernel"),
                "traffic_unit": "L2 fabric-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, KB units, separate rocprofv3 --pmc passes): the counters sit "
                                "between L2 and the fabric and INCLUDE Infinity-Cache hits, i.e. an upper bound of the HBM bytes",
                "traffic_source": traffic_source(args, n_frames),
                "launches": prof["gmm_launches"], "chunks_per_step": prof["gmm_launches"] / steps, "ms_per_step": ms,
                "avg_launch_ms": prof["refine_ms"] / launches, "frames_per_launch": n_frames * steps / launches,
                "flops_per_frame": 4.0 * D * S, "dtype": "f64 unfused add/mul",
                "note": "dominant kernel of the step; it is bound by the FP64 vector pipe (no MFMA, HBM traffic hidden), which "
                        "the hbm|mfma enum cannot name; the MFMA-bound prefilter kernel is under roofline_prefilter, the dense "
                        "FP64 MFMA path (--kernel mfma) reaches 0.86 of the 78.6 TF matrix peak"}
    gmm_s = prof["gmm_ms"] * 1e-3
    achieved = prof["gmm_flops"] / gmm_s / 1e12 if gmm_s > 0 else 0.0
    return {"kernel": "gmm_mfma_kernel" if args.kernel == "mfma" else "gmm_exact_kernel", "bound": "mfma",
            "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            "traffic": pmc_traffic(args, n_frames, "gmm_mfma_kernel") if args.kernel == "mfma" else None,
            "traffic_unit": "L2 fabric-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE; includes Infinity-Cache hits)",
            "launches": prof["gmm_launches"], "chunks_per_step": prof["gmm_launches"] / steps, "avg_launch_ms": prof["gmm_ms"] / launches,
            "frames_per_launch": n_frames * steps / launches, "flops_per_frame": 4.0 * D * S * args.mix}


def padded_slots(word_off, automaton, silence_idx, silence_state):
    """Trellis positions after the fast decoder's type sort: every (kind, silence flags) group is padded to whole 64-slot
    chunks (sr_lexicon_create: build of the type-sorted net)."""
    import collections
    groups = collections.Counter()
    for w in range(len(word_off) - 1):
        b, n = int(word_off[w]), int(word_off[w + 1] - word_off[w])
        first = int(automaton[b])
        for k in range(n):
            kind = (1 if n == 1 else 0) if k == 0 else (3 if n == 2 else 2) if k == 1 else (5 if k == n - 1 else 4)
            st = int(automaton[b + k])
            groups[kind | (int(st == silence_state) << 3) | (int(w == silence_idx) << 4) | (int(first == silence_state) << 5)] += 1
    return sum((c + 63) // 64 * 64 for c in groups.values())


def decode_geometry(P, n_utts):
    """threads x slots per thread of decode_fast_kernel for P type-padded trellis positions in a launch of n_utts utterances
    (launch_decode_fast: the widest workgroup when there is at most one utterance per CU, narrower ones for throughput);
    None beyond the LDS kernels' 8192 slots (decode_big_kernel)."""
    if P > 8192:
        return None
    small = ((64, "64, 1"), (256, "256, 1"), (1024, "1024, 1"), (2048, "1024, 2")) if n_utts <= 256 else \
            ((64, "64, 1"), (256, "64, 4"), (1024, "256, 4"), (2048, "512, 4"))
    for limit, geom in small + ((4096, "1024, 4"), (8192, "1024, 8")):
        if P <= limit:
            return geom
    return None


def search_report(args, prof, S, P, P_padded, n_frames, n_utts, network):
    """The Viterbi step against SURVEY 8(d)'s HBM model (8*S + 4*P bytes per frame).  `network` = sr_lexicon_describe: which
    search kernel the library runs on this lexicon.  The word-per-lane kernel (short-word lexica: all of SURVEY 8d's) keeps the
    hypotheses in registers and reads nothing but the score rows: its time is the rows' HBM fetch at one row in flight per
    workgroup, two workgroups per CU; the slot-per-lane kernel is bound by its per-frame dependency chain."""
    ms = prof["search_ms"] / max(1, args.steps)
    algorithmic = prof["search_bytes"] / (prof["search_ms"] * 1e-3) / 1e9 if prof["search_ms"] > 0 else 0.0
    words = network.startswith("words")
    traffic = pmc_traffic(args, n_frames, "decode_words_kernel" if words else "decode_fast_kernel")
    geom = decode_geometry(P_padded, n_utts)
    if words:
        _, nw, _, nt, _, L = network.split()[:6]
        kernel = (f"decode_words_kernel<{nw}, {L}, {'true' if network.endswith('general') else 'false'}> x {nt} lanes (one lane per word, hypotheses in registers, "
                  "score rows staged in LDS by LDS-DMA, one barrier per frame; + decode_kernel replay of flagged utterances)")
        bound = "HBM fetch of the score rows (the only memory traffic) at one row in flight per workgroup, two workgroups per CU; then the per-frame barrier"
    elif geom:
        kernel = f"decode_fast_kernel<{geom}> (score rows staged in LDS by LDS-DMA when they fit; + decode_kernel replay of flagged utterances)"
        bound = "latency of the per-frame dependency chain (two barriers per frame, one workgroup of 16 waves per CU); not HBM"
    else:
        kernel = "decode_big_kernel<1024> (hypotheses in device memory: more than 8192 type-padded slots)"
        bound = "device-memory round trips of the hypothesis arrays"
    return {
        "kernel": kernel,
        "network": network,
        "ms_per_step": ms,
        "bound": bound,
        # ONE formula: the library's own count (sr_profile.search_bytes = (8 S + 4 P) x frames, P = the search's trellis positions --
        # for the bigram search the words plus their silence copies) over what it was counted on
        "hbm_model": {"bytes_per_frame": prof["search_bytes"] / max(1.0, float(prof["frames"])), "formula": "8 S + 4 P (SURVEY 8d)",
                      "achieved_GBps": algorithmic, "peak_GBps": 8000.0, "frac": algorithmic / 8000.0},
        "fabric_measured_bytes_per_launch": traffic,  # (L2 fabric-side counters: Infinity-Cache hits included)
        "fabric_measured_GBps": traffic / (ms * 1e-3) / 1e9 if traffic and ms > 0 else None,
    }


def prefilter_report(args, prof, n_frames, D, S):
    launches = max(1, prof["gmm_launches"])
    steps = max(1, args.steps)
    p_ms, g_ms = prof["prefilter_ms"] / steps, prof["gmm_ms"] / steps   # per STEP: the flops below are one step's (all chunks)
    k = 32 * ((2 * D + 3 + 31) // 32)
    cs = 1 if args.mix <= 32 else 2 if args.mix <= 64 else 4   # a mixture of more than 32 densities is cut into 2 or 4 chunks of 32 (pseudo-states)
    p_flops = 2.0 * k * (32 * 4 * ((S * cs + 3) // 4)) * n_frames  # executed: one fp16 product, K and (pseudo-)states padded
    p_useful = 2.0 * (2 * D + 3) * S * args.mix * n_frames     # useful: K = 2 D + 3 per real density
    dense = 4.0 * D * S * args.mix * n_frames
    # matrix-pipe busy share from the committed counters (same workload, same kernel sources): SQ_VALU_MFMA_BUSY_CYCLES over
    # 128 pipe-cycles per GRBM_GUI_ACTIVE cycle (4 SIMDs x 256 CUs / 8 XCDs: the counter sums the XCDs)
    busy = None
    z = pmc_summary(args, n_frames)
    if z is not None:
        for name, k in z.get("kernels", {}).items():
            p = k.get("pmc_mean_per_dispatch", {})
            if name.startswith("gmm_prefilter16_kernel") and p.get("GRBM_GUI_ACTIVE"):
                busy = p.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 128.0 / p["GRBM_GUI_ACTIVE"]
    return {
        "roofline_prefilter": {"kernel": "gmm_prefilter16_kernel", "bound": "mfma", "achieved": p_flops / (p_ms * 1e-3) / 1e12,
                               "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": p_flops / (p_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
                               "frac_useful": p_useful / (p_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
                               "ms_per_step": p_ms, "avg_launch_ms": prof["prefilter_ms"] / launches, "chunks_per_step": prof["gmm_launches"] / steps,
                               "dtype": "f16 x f16 -> f32", "includes": "feature transpose (0.04 ms)",
                               "mfma_pipe_busy": busy, "mfma_pipe_busy_source": traffic_source(args, n_frames),
                               "note": "frac counts the padded K = 96 that the MFMAs execute, frac_useful only K = 81; the matrix pipe is "
                                       "busy about half of the cycles (mfma_pipe_busy, from the PMC summary), the rest is the mask "
                                       "epilogue's vector issue (DESIGN 4.1)"},
        "gmm_step": {"ms": g_ms, "chunks_per_step": prof["gmm_launches"] / steps, "dense_fp64_flops": dense, "dense_fp64_equiv_tflops": dense / (g_ms * 1e-3) / 1e12,
                     "vs_fp64_mfma_peak": dense / (g_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                     "densities_refined_per_pair": prof["refined_densities"] / max(1, prof["refined_pairs"]), "of": args.mix,
                     "note": "scores bit-identical to the reference; SURVEY 8(d)'s 4*D*C flops per frame are not executed in "
                             "FP64 any more, so the dense-equivalent rate exceeds the FP64 peak"},
    }


def build_identity():
    """Commit the library was built from (speechrecognition_amd/build.py records it: the GPU box has no .git) + the hash of
    the device sources in this tree."""
    from speechrecognition_amd import build as b
    info = b.build_info()
    return {"git_head": (info.get("git_head", "unknown") + ("+dirty" if info.get("dirty") else "")), "kernel_sources_sha16": kernel_sources_sha16()}


def kernel_sources_sha16():
    """sha256 over the device sources, as tools/summarize_profile.py stamps a profile with."""
    import hashlib
    csrc = os.path.join(ROOT, "speechrecognition_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


PMC_SUMMARIES = {"prefilter": ("r5_prefilter_summary.json", "r4_prefilter_summary.json", "r3_prefilter_summary.json", "r2_prefilter_summary.json"), "mfma": ("r1_mfma_summary.json",)}


def pmc_summary(args, n_frames):
    """The PMC summary under profiles/ (tools/profile_bench.sh; rocprofv3 cannot run inside the timed process) that belongs to
    THIS workload and to THESE kernel sources, or None: counters taken before a kernel changed are not reported."""
    for name in PMC_SUMMARIES.get(args.kernel, ()):
        try:
            z = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if z.get("workload_frames_per_launch") != n_frames or args.words != 1333 or args.mix != 32:
            continue
        if z.get("kernel_sources_sha16") != kernel_sources_sha16():
            continue
        z["_file"] = "profiles/" + name
        return z
    return None


def pmc_traffic(args, n_frames, kernel="gmm_mfma_kernel"):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ -- only when they were taken on this
    very workload with the kernel sources of this tree (the summary carries their hash and commit)."""
    z = pmc_summary(args, n_frames)
    if z is None:
        return None
    for name, k in z.get("kernels", {}).items():
        if name.startswith(kernel):
            return k.get("hbm_bytes_per_launch_corrected")
    return None


def traffic_source(args, n_frames):
    z = pmc_summary(args, n_frames)
    if z is None:
        return "none: no PMC summary under profiles/ matches this workload and these kernel sources (sha %s)" % kernel_sources_sha16()
    return {"file": z["_file"], "git_head": z.get("git_head"), "kernel_sources_sha16": z.get("kernel_sources_sha16")}


def usable_cores():
    """Threads we may really use: the cgroup CPU quota if there is one, else the affinity mask, and never
    more than 16 (the GPU box gives one GPU's job a 16-CPU share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_model():
    """(model name, logical CPUs of the host) from /proc/cpuinfo -- SURVEY 8(d): "core count and CPU model printed"."""
    name, n = "unknown", 0
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                n += 1
                name = line.split(":", 1)[1].strip()
    except OSError:
        pass
    return name, n


def cpu_baseline(args, mixset_path, lex, tdp, wp, feats, frame_off, gpu_words, gpu_woff):
    """The CPU oracle's utterance loop (the reference's timed region, Recognizer.cpp:45-80) on a bounded sample of the same
    batch, SURVEY 8(d)'s two legs: (i) ONE thread -- the shortest utterance, which also yields the reference's lazy-scoring
    fraction (scorer calls / (frames x states): its am_cache scores only states a live hypothesis touches,
    Recognizer.cpp:123,148-151,178-181) -- and (ii) all usable cores with the reference's strategy (OpenMP
    schedule(dynamic) over utterances, :46).  The sample's words are checked against the GPU's."""
    from oracle import pyoracle

    cores = usable_cores()
    model_name, host_cpus = cpu_model()
    orc = pyoracle.Oracle(mixset_path, 39, lex, tdp=tdp, am_threshold=args.beam, word_penalty=wp, max_approx=not args.sum_mode)
    lens = np.diff(frame_off.astype(np.int64))
    u0 = int(np.argmin(lens))
    f0 = feats[int(frame_off[u0]):int(frame_off[u0 + 1])]
    t = time.perf_counter()
    w0 = orc.decode(f0)
    secs1 = time.perf_counter() - t
    per_frame = secs1 / max(1, len(f0))
    lazy = orc.last_n_scored / float(max(1, len(f0)) * lex.n_states)
    one = {"value": len(f0) / secs1, "unit": "frames/s", "cores": 1,
           "sample": f"utterance {u0} (the shortest: {len(f0)} frames), {secs1:.1f} s wall",
           "words_match_gpu": bool(np.array_equal(w0, gpu_words[int(gpu_woff[u0]):int(gpu_woff[u0 + 1])]))}
    budget_frames = args.cpu_seconds * cores / max(per_frame, 1e-9)
    n = int(np.searchsorted(frame_off[1:].astype(np.float64), budget_frames)) + 1
    n = max(min(n, len(lens)), min(cores, len(lens)))
    sub_off = frame_off[: n + 1].copy()
    sub_feats = feats[: int(sub_off[-1])]
    words, woff, secs = orc.recognize_batch(sub_feats, sub_off, n_threads=cores)
    match = bool(np.array_equal(words, gpu_words[: int(gpu_woff[n])]) and np.array_equal(woff, gpu_woff[: n + 1]))
    fps = float(sub_off[-1]) / secs
    orc.close()
    return {
        "value": fps,
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n} of {len(lens)} utterances ({int(sub_off[-1])} frames), lazy scoring + beam {args.beam:g}, "
                  f"OpenMP schedule(dynamic) over utterances, {secs:.1f} s wall",
        "words_match_gpu": match,
        "port_vs_reference": "the port (oracle/sr_oracle.c) is FASTER than the reference compiled from its own sources in the build container: 705 against "
                             "581 frames/s on the same 791 words (tools/ref_vs_oracle_speed.py, 21 %; 604 / 546 in round 3) -- a GPU / CPU ratio "
                             "read off this line understates the ratio against the reference itself",
        "cpu_model": model_name,
        "host_logical_cpus": host_cpus,
        "one_thread": one,
        "reference_equivalent": {
            "lazy_scored_fraction": lazy,
            "note": "share of the (frame, state) pairs the reference's lazy am_cache scores at this beam (measured by the oracle on the "
                    "one-thread utterance); the GPU scores every pair, so its reference-equivalent rate is `value` of this line "
                    "in frames/s either way and its dense density-evaluation rate is 1/fraction of the reference-equivalent one",
        },
    }


def cpu_baseline_bigram(args, mixset_path, lex, lm, bg_tdp, feats, frame_off, gpu_words, gpu_woff):
    """Dense scoring (OpenMP over frames) + the bigram search restatement on the shortest utterances that fit the CPU
    budget; words checked against the GPU's."""
    from oracle import pyoracle

    cores = usable_cores()
    orc = pyoracle.Oracle(mixset_path, 39, lex)
    word_off, mixtures, _ = lex.flatten()
    lens = np.diff(frame_off.astype(np.int64))
    order = np.argsort(lens, kind="stable")
    t0 = time.perf_counter()
    done, frames, match = 0, 0, True
    for u in order:
        x = feats[int(frame_off[u]):int(frame_off[u + 1])]
        dense = orc.score_matrix(x, n_threads=cores)
        w, _, _ = pyoracle.bigram_decode(dense, word_off, mixtures, lex.silence_idx, lm, bg_tdp, args.beam, pyoracle.FLT_MAX)
        match = match and bool(np.array_equal(w, gpu_words[int(gpu_woff[u]):int(gpu_woff[u + 1])]))
        done += 1
        frames += len(x)
        if time.perf_counter() - t0 > args.cpu_seconds:
            break
    secs = time.perf_counter() - t0
    orc.close()
    return {"value": frames / secs, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{done} shortest of {len(lens)} utterances ({frames} frames), dense scoring on {cores} threads + "
                      f"sequential bigram search, {secs:.1f} s wall", "words_match_gpu": match}


if __name__ == "__main__":
    main()
