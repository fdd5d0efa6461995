#!/usr/bin/env python3
"""cProfile of bench.py's host side (run on the GPU box): python tools/prof_host.py"""
import cProfile
import io
import os
import pstats
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ['bench.py', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-extras'] + sys.argv[1:]
sys.path.insert(0, ROOT)
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
