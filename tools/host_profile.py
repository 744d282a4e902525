#!/usr/bin/env python3
"""cProfile of the host side of one chain of tools/chain_profile.py (development tool):
    python tools/host_profile.py llava-crop generic
The loops of the generic chains run at the host's pace; this shows where the host's time goes."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import chain_profile  # noqa: E402

if __name__ == "__main__":
    pr = cProfile.Profile()
    pr.enable()
    chain_profile.main()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
