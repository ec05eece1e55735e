set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/b_def.jsonl 2> gpurun_out/r2a/b_def.err && \
python bench.py --spawn --no-cpu-baseline > gpurun_out/r2a/b_spawn1.jsonl 2> gpurun_out/r2a/b_spawn1.err && \
GEV_BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r2a/b_n2.jsonl 2> gpurun_out/r2a/b_n2.err && \
GEV_BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --migration-rate 0.01 > gpurun_out/r2a/b_n2mig.jsonl 2> gpurun_out/r2a/b_n2mig.err
echo "bench rc=$?"
tail -3 gpurun_out/r2a/pytest.log
