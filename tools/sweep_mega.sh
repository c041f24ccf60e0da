for w in 4 6 8; do echo "win=$w"; MRBF_MEGA_DEBUG=1 MRBF_MEGA_WIN=$w timeout -k 10 200 python tools/mega_check.py 4096,8192,12288,16384 3 2 2>&1 | grep "n=\|mega"; done
