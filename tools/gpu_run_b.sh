set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests_b.log 2>&1 || { tail -40 gpurun_out/r3_tests_b.log; exit 1; }
tail -3 gpurun_out/r3_tests_b.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-dense-mfma > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3_bench_b.json'))
print('step', d['ms_per_step'], 'R', d['roofline']['avg_launch_ms'], 'P', d['roofline_prefilter']['avg_launch_ms'], 'search', d['search']['ms_per_step'], 'words', d['recognised_words_rank0'])"
