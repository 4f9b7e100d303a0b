#!/usr/bin/env python3
"""cProfile of the end-to-end run of tools/e2e_configs.py (host-side costs)."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import e2e_configs  # noqa: E402

sys.argv = ['e2e_configs.py'] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
e2e_configs.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
