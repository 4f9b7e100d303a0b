#!/usr/bin/env python3
"""cProfile of tools/e2e_twopass.py (host-side costs of the two-pass workflow)."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import e2e_twopass  # noqa: E402

sys.argv = ['e2e_twopass.py'] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
e2e_twopass.main()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
