O=gpurun_out/r05a; mkdir -p $O
timeout 400 python tools/bwd_variant_check.py > $O/variant_check.txt 2>&1; echo "check rc=$?" >> $O/variant_check.txt
timeout 400 python tools/bwd_ab.py --rounds 5 --steps 50 --cases all,gg,g --variants 0,10,12,13 > $O/bwd_ab_4k.txt 2>&1
timeout 300 python tools/bwd_ab.py --workload 1080p --rounds 5 --steps 100 --cases all,gg,g --variants 0,10,12,13 > $O/bwd_ab_1080p.txt 2>&1
timeout 300 python tools/bwd_ab.py --smooth --rounds 3 --steps 50 --cases all,g --variants 0,10 > $O/bwd_ab_4k_smooth.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_rccl.py -x -q > $O/rccl.txt 2>&1
for i in 1 2 3; do HDRNET_BENCH_TRACE=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined; done > $O/bench_k20.txt 2>&1
timeout 300 python bench.py --workload train_1080p_b4 --force-collective --steps 50 --warmup 10 > $O/bench_train_forced.txt 2>&1
tail -3 $O/variant_check.txt; cat $O/bwd_ab_4k.txt | tail -14; tail -5 $O/rccl.txt
