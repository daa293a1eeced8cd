#!/usr/bin/env python3
"""Regenerate EVERY fixture under tests/golden/ from the reference into an EMPTY scratch directory -- twice -- and
compare every file with the committed one byte for byte.  All ten generators run, in dependency order: gen_golden.py
first (it writes the lens states and fixture f8 that gen_golden_handoff.py reads back), then the other nine, which are
independent of each other and run side by side (each is single-threaded: torch.set_num_threads(1)).  Nothing is copied
from tests/golden/; the run-to-run-unstable scalars of the reference (paraxial pupils, hfov, foclen, fnum) come from
oracle/frozen_lens_scalars.json, and each generator asserts that this run's fresh values lie within the reference's own
spread of them.  Build container only (imports /root/reference).  TEST INFRASTRUCTURE.

Usage: python oracle/check_regenerable.py [--runs 2] [--jobs 8]     (output kept in profiles/rNN/regenerable.txt)"""
import argparse
import filecmp
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")
FIRST = "gen_golden.py"
REST = ["gen_golden_variant.py", "gen_golden_pupil.py", "gen_golden_splat_fuzz.py", "gen_golden_handoff.py",
        "gen_golden_boundary.py", "gen_golden_callers.py", "gen_golden_analysis.py", "gen_golden_psfnet.py",
        "gen_golden_dfdp.py"]


def regenerate(out_dir, jobs=8):
    """Run the ten generators into out_dir (empty).  -> {script: seconds}; raises with the script's output on failure."""
    took = {}

    def start(script):
        return script, time.time(), subprocess.Popen([sys.executable, os.path.join(HERE, script), "--out", out_dir],
                                                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    def finish(script, t0, proc):
        out, _ = proc.communicate()
        took[script] = time.time() - t0
        if proc.returncode != 0:
            raise RuntimeError(f"{script} failed ({proc.returncode}):\n{out[-3000:]}")
    finish(*start(FIRST))
    todo, running = list(REST), []
    todo.sort(key=lambda s: s != "gen_golden_analysis.py")          # the long one (analysis_rms: 2.4 minutes) first
    while todo or running:
        while todo and len(running) < max(1, jobs):
            running.append(start(todo.pop(0)))
        done = [r for r in running if r[2].poll() is not None]
        if not done:
            time.sleep(0.2)
            continue
        for r in done:
            running.remove(r)
            finish(*r)
    return took


def compare(out_dir):
    """-> (files written, byte-identical to the committed ones, committed files nobody wrote)."""
    names = sorted(os.listdir(out_dir))
    same = [n for n in names if os.path.exists(os.path.join(GOLDEN, n))
            and filecmp.cmp(os.path.join(out_dir, n), os.path.join(GOLDEN, n), shallow=False)]
    missing = sorted(set(os.listdir(GOLDEN)) - set(names))
    return names, same, missing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--jobs", type=int, default=min(8, os.cpu_count() or 1))
    args = ap.parse_args()
    bad = 0
    for run in range(1, args.runs + 1):
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.time()
            took = regenerate(tmp, args.jobs)
            names, same, missing = compare(tmp)
            print(f"run {run}: {len(took)} generators, {len(names)} files written into an empty directory in {time.time() - t0:.0f} s, "
                  f"{len(same)} byte-identical to tests/golden/; different: {sorted(set(names) - set(same))}; "
                  f"committed but not regenerated: {missing}")
            print("         per generator [s]: " + ", ".join(f"{k[:-3]} {v:.0f}" for k, v in took.items()))
            bad += len(names) - len(same) + len(missing)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
