#!/usr/bin/env python3
"""Regenerate the fixtures of oracle/gen_golden.py from the reference into a scratch directory -- twice -- and compare
every file with the committed tests/golden/ byte for byte.  Build container only (imports /root/reference).
TEST INFRASTRUCTURE.  Usage: python oracle/check_regenerable.py  (output kept in profiles/rNN/regenerable.txt)"""
import filecmp
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")
bad = 0
for run in (1, 2):
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.time()
        subprocess.check_call([sys.executable, os.path.join(HERE, "gen_golden.py"), "--out", tmp],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        names = sorted(os.listdir(tmp))
        same = [n for n in names if filecmp.cmp(os.path.join(tmp, n), os.path.join(GOLDEN, n), shallow=False)]
        print(f"run {run}: {len(names)} files written in {time.time() - t0:.0f} s, {len(same)} byte-identical to tests/golden/; "
              f"different: {sorted(set(names) - set(same))}")
        bad += len(names) - len(same)
sys.exit(1 if bad else 0)
