set -x
mkdir -p gpurun_out
for seed in 61 62; do python tools/fuzz_vs_oracle.py $seed 2000 2>&1 | sed "s/^/seed $seed: /" ; done > gpurun_out/r6_fuzz.txt 2>&1
tail -3 gpurun_out/r6_fuzz.txt
for c in 3 2 4 5; do python bench.py --config $c > gpurun_out/r6_bench_c$c.json 2> gpurun_out/r6_bench_c$c.err; tail -c 300 gpurun_out/r6_bench_c$c.json; done
