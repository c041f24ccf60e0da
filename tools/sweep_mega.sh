for look in 0 1; do for ded in 32 64 128; do echo "look=$look ded=$ded"; MRBF_MEGA_LOOK=$look MRBF_MEGA_DEDICATED=$ded timeout -k 10 100 python tools/mega_check.py 8192 3 3 | tail -1; done; done
for ch in 8 16; do echo "chain=$ch"; MRBF_MEGA_CHAIN=$ch timeout -k 10 100 python tools/mega_check.py 8192 3 3 | tail -1; done
