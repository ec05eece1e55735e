set -x
O=gpurun_out/r2g; mkdir -p $O
for g in 128 256 384 640 1024; do
GEV_SAMPLE_GRID=$g python bench.py --steps 10 --warmup 4 --no-cpu-baseline --isolated-steps 0 --nchr 11 --n-ind 125000 --n-loci 227000 > $O/s11_sg$g.jsonl 2> $O/s11_sg$g.err
done
for g in 128 256 384 640 1024; do
GEV_SAMPLE_GRID=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline --isolated-steps 0 > $O/c2_sg$g.jsonl 2> $O/c2_sg$g.err
done
GEV_OVERLAP=2 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --isolated-steps 0 --nchr 11 --n-ind 125000 --n-loci 227000 > $O/s11_ovl2.jsonl 2> $O/s11_ovl2.err
GEV_STITCH_WG_PER_CU=6 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --isolated-steps 0 --nchr 11 --n-ind 125000 --n-loci 227000 > $O/s11_occ6.jsonl 2> $O/s11_occ6.err
