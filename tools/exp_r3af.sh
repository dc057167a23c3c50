#!/bin/bash
# Host submission time per frame against GPU time per frame: N = 1 (C3) and a rank's share at 8 emulated ranks, without / with the exchange.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>&1 >/dev/null | grep "host submission" | cut -c1-300
TR_NO_GATHER=1 python3 bench.py --emulate-ranks 8 --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>&1 >/dev/null | grep "host submission" | cut -c1-300
python3 bench.py --emulate-ranks 8 --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>&1 >/dev/null | grep "host submission" | cut -c1-300
TRHIP_HOST_PROFILE=1 TR_NO_GATHER=1 python3 bench.py --emulate-ranks 8 --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>&1 >/dev/null | grep -i -A30 "host profile\|us per" | head -40
