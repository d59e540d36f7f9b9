"""Scratch: cProfile of the continuous-learning loop (host side)."""
import sys, os, cProfile, pstats, io
sys.argv = [sys.argv[0], "8000"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pr = cProfile.Profile()
pr.enable()
exec(open(os.path.join(ROOT, "tools", "add_examples_probe.py")).read())
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
