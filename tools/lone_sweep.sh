for d in 16 32 64; do for c in 128 256 512 1024 2048; do for L in 0 1; do BFHIP_LONE=$L timeout 120 python tools/lone_rate.py gauss $d $c 150 2>&1 | tail -1; done; done; done
for c in 256 512 1024; do for L in 0 1; do BFHIP_LONE=$L timeout 120 python tools/lone_rate.py funnel 64 $c 100 2>&1 | tail -1; done; done
