#!/bin/bash
# tools/groupsweep.sh -- C2 box with 1, 2 (default for 4K), 4, 8 frame groups, three rounds
for r in 1 2 3; do for g in 2 4 8 1; do python bench.py --no-extra --no-cpu-baseline --option groups=$g 2>/dev/null | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print('C2 groups $g', j['roofline']['kernel_ms_per_step'], j['roofline']['frac'])"; done; done | sort | awk '{k=$1" "$2" "$3; s[k]=s[k]" "$4} END{for(k in s) print k, s[k]}' | sort
