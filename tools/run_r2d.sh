set -x
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
grep "gev_rank_f64" $O/pytest.log
./tools/stitch_bench > $O/stitch_bench_c2.txt 2>&1
./tools/stitch_bench 125000 227000 > $O/stitch_bench_s11.txt 2>&1
python bench.py --no-cpu-baseline > $O/b_def.jsonl 2> $O/b_def.err
GEV_STITCH_MODE=2 python bench.py --no-cpu-baseline > $O/b_def_mode2.jsonl 2> $O/b_def_mode2.err
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --nchr 11 --n-ind 125000 --n-loci 227000 > $O/b_shard11.jsonl 2> $O/b_shard11.err
python tools/cli_timing.py --exe gpu --gens 3 > $O/cli_gpu.json 2> $O/cli_gpu.err
cat $O/stitch_bench_c2.txt
