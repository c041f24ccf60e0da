for wb in 0 1 2 4 8; do echo "wbias=$wb"; MRBF_MEGA_WBIAS=$wb timeout -k 10 200 python tools/mega_check.py 8192,16384 3 3 2>&1 | grep "n="; done
