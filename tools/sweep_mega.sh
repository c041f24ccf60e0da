for fw in 4 2 1; do echo "first_window=$fw"; MRBF_MEGA_FIRST_WINDOW=$fw timeout -k 10 100 python tools/mega_check.py 2048,4096,8192 3 4 | grep "n="; done
