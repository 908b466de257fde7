#!/bin/bash
# developer sweep: FPS launch time over batch size and K (prologue / early rounds / late rounds split)
for B in ${BS:-1 256 300}; do timeout -k 10 120 python tools/fps_probe.py --frames $B --K ${KS:-2 16 64 128 256 512 1024} --algo ${ALGO:-2} --reps 5 2>&1 | grep -E "K=|frames" ; done
