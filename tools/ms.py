#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> 'ms_per_step kernel_ms value'."""
import json
import sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r["ms_per_step"], r["roofline"]["kernel_ms"], r["value"])
