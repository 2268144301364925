O=gpurun_out/r05b; mkdir -p $O
timeout 400 python tools/bwd_variant_check.py --variants 10 > $O/variant_check.txt 2>&1; echo "check rc=$?" >> $O/variant_check.txt
timeout 400 python tools/bwd_ab.py --rounds 5 --steps 50 --cases all,gg,g --variants 0,10 > $O/bwd_ab_4k.txt 2>&1
timeout 300 python tools/bwd_ab.py --workload 1080p --rounds 5 --steps 100 --cases all,gg,g --variants 0,10 > $O/bwd_ab_1080p.txt 2>&1
tail -4 $O/variant_check.txt; grep case $O/bwd_ab_4k.txt; grep case $O/bwd_ab_1080p.txt
