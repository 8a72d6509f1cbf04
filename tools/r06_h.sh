#!/bin/bash
timeout 300 python tools/bench_epoch.py 100 2>&1 | tail -1
MVAE_NO_PAD_ROWS=1 timeout 300 python tools/bench_epoch.py 100 2>&1 | tail -1
timeout 300 python tools/bench_epoch.py 128 2>&1 | tail -1
