# sweep of the first-round start shifts of k_stream_wide: GTCRN_STAGGER="phases,unit" (unit = sleeps of 512 clocks per shift)
for st in ${@:-16,12 32,6 64,3 128,2 256,1 32,8 32,4 64,2 8,0}; do export GTCRN_STAGGER=$st; echo "stagger $st"; timeout 200 python tools/stream_form_ab.py 3 3 8192 16384 65536 2>&1 | tail -1; done
