#!/bin/bash
# A/B of two builds of the library in one GPU session: alternating bench.py runs (img/s), reference first.
# usage: ab_step.sh <reference .so> [rounds]
ref=$1; rounds=${2:-2}
for i in $(seq $rounds); do
  SFCVIT_LIB=$ref python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ref ', d['value'], d['ms_per_step'])"
  python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'])"
done
