#!/usr/bin/env python3
"""Experiment: how does k_match_global_v2 react to fewer resident workgroups per CU?  (SF_MATCH_LDS_PAD
inflates its dynamic LDS request: 160 KB / (22.6 KB + pad) workgroups fit.)  Diagnostic only."""
import os, subprocess, sys
for pad in (0, 10000, 18000, 31000, 58000):
    env = dict(os.environ, SF_MATCH_LDS_PAD=str(pad))
    out = subprocess.run([sys.executable, "tools/ab_match.py", "10000", "500", "32", "12256"], env=env,
                         capture_output=True, text=True)
    tail = [l for l in out.stdout.splitlines() if l.strip()][-2:]
    print("pad %6d B -> %d workgroups/CU by LDS : %s" % (pad, 160 * 1024 // (22592 + pad), tail), flush=True)
