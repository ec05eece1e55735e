set -x
O=gpurun_out/r2h; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/b_def.jsonl 2> $O/b_def.err
python bench.py --no-cpu-baseline --plane-less > $O/b_planeless.jsonl 2> $O/b_planeless.err
python tools/cli_timing.py --exe gpu --gens 3 > $O/cli_gpu.json 2> $O/cli_gpu.err
python tools/cli_timing.py --exe gpu --gens 3 --assortative > $O/cli_gpu_am.json 2> $O/cli_gpu_am.err
