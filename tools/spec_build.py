#!/usr/bin/env python3
"""Per-prescription specialisation of the fused PSF kernel (experimental).

  python tools/spec_build.py rf50mm [--wvln 0.589] [--tag spec_rf50mm]

writes build/<tag>.hpp (sdirt_emit_spec: the constants of the lens tables as literals, the
surface loop unrolled) and compiles build/libsdirt_dp_<tag>.so = the product library with the
fused kernel's trace replaced by it.  The library is valid for THIS lens at THIS wavelength only:
  SDIRT_AMD_LIB=build/libsdirt_dp_<tag>.so python tools/kbench.py
Runs without a GPU (the tables are built on the host)."""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lens")
    ap.add_argument("--wvln", type=float, default=0.589)
    ap.add_argument("--tag", default=None)
    args = ap.parse_args()
    tag = args.tag or f"spec_{args.lens}"
    from sdirt_amd import _lib
    from sdirt_amd.basics import DEFAULT_WAVE
    from conftest import load_state, make_lens
    lens = make_lens(args.lens, "cpu", load_state(args.lens))
    K = len(lens.surfaces)
    arr_t = _lib.SurfaceDesc * K
    prim = arr_t(*[s.desc(args.wvln) for s in lens.surfaces])
    cen = arr_t(*[s.desc(DEFAULT_WAVE) for s in lens.surfaces])
    h = _lib.lib()
    n = h.sdirt_emit_spec(prim, cen, K, None, 0)
    assert n > 0, n
    buf = C.create_string_buffer(n)
    assert h.sdirt_emit_spec(prim, cen, K, buf, n) == n
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    hdr = os.path.join(ROOT, "build", f"{tag}.hpp")
    open(hdr, "wb").write(buf.value)
    print(f"wrote {hdr} ({n} bytes, {K} surfaces)")
    rc = subprocess.call([os.path.join(ROOT, "tools", "build_variant.sh"), tag, f'-DSDIRT_SPEC_HEADER="{hdr}"'])
    sys.exit(rc)


if __name__ == "__main__":
    main()
