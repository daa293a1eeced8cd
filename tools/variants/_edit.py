"""Helpers for the variant edit scripts: exact-text replacement that fails loudly."""
import os
import sys


def sub(root, fname, old, new, count=1):
    p = os.path.join(root, fname)
    s = open(p).read()
    if s.count(old) < 1 or (count and s.count(old) != count):
        sys.exit(f"{fname}: expected {count} occurrence(s) of {old[:60]!r}, found {s.count(old)}")
    open(p, "w").write(s.replace(old, new))
