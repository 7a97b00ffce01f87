set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
cut -c1-300 gpurun_out/r3_bench.json
timeout -k 10 300 python tools/time_single_utt.py > gpurun_out/r3_single_utterance.txt 2>&1 || tail -5 gpurun_out/r3_single_utterance.txt
cat gpurun_out/r3_single_utterance.txt
timeout -k 10 300 python tools/time_training_iter.py > gpurun_out/r3_training_iteration.txt 2>&1 || tail -5 gpurun_out/r3_training_iteration.txt
tail -5 gpurun_out/r3_training_iteration.txt
timeout -k 10 600 python bench.py --config cfg5 --steps 3 --warmup 1 > gpurun_out/r3_bench_bigram_cfg5.json 2> gpurun_out/r3_bench_cfg5.err
cut -c1-200 gpurun_out/r3_bench_bigram_cfg5.json
timeout -k 10 600 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-dense-mfma > gpurun_out/r3_bench_cfg4_one_gpu.json 2> gpurun_out/r3_bench_cfg4.err
cut -c1-200 gpurun_out/r3_bench_cfg4_one_gpu.json
timeout -k 10 600 python bench.py --gpus 2 --dist-backend gloo --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_2rank_gloo.json 2> gpurun_out/r3_bench_2rank.err || tail -5 gpurun_out/r3_bench_2rank.err
cut -c1-300 gpurun_out/r3_bench_2rank_gloo.json
