for sl in 2 3; do for ded in 32 64; do echo "slack=$sl ded=$ded"; MRBF_MEGA_SLACK=$sl MRBF_MEGA_DEDICATED=$ded timeout -k 10 100 python tools/mega_check.py 8192 3 3 | tail -1; done; done
