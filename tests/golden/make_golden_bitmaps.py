"""Generates tests/golden/bitmap_*.bmp with the UNMODIFIED reference tool
(oracle/_ref/DiscreteFieldToBitmap = the reference's cmd/discrete_field_to_bitmap/main.cpp +
bmp_file.cpp, built by `make -C oracle ref`; runs in the build container only) from the committed
golden .cdf/.cdm files.  The reference writes biSizeImage (file offsets 34..37) from an
uninitialised variable (bmp_file.cpp:85-101: the header is written before the field is
assigned), so those four bytes differ from run to run; they are zeroed in the fixtures.

Run:  python tests/golden/make_golden_bitmaps.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import dgtest as T  # noqa: E402

# name -> (input file, tool options); tests/test_bitmap.py reads this table
CASES = {
    "torus16_xy": ("torus_16_16_6.cdf", ["-s", "48"]),
    "torus16_xz_d03": ("torus_16_16_6.cdf", ["-s", "50", "-p", "xz", "-d", "0.3"]),
    "torus9_reduced_yz_rs": ("torus_9_14_6_reduced_0p08.cdf", ["-s", "41", "-p", "yz", "-c", "rs", "-d", "-0.1"]),
    "torus16_density_yx": ("torus_16_16_6_density_reduced.cdm", ["-f", "1", "-s", "33", "-p", "yx", "-c", "rs", "-d", "0.45"]),
    "box_zx": ("box.cdf", ["-s", "31", "-p", "zx", "-d", "0.5"]),
}


def main():
    exe = os.path.join(T.ROOT, "oracle", "_ref", "DiscreteFieldToBitmap")
    assert os.path.exists(exe), "build oracle/_ref first: make -C oracle ref"
    for name, (src, opts) in CASES.items():
        out = os.path.join(HERE, "bitmap_%s.bmp" % name)
        subprocess.check_call([exe] + opts + ["-o", out, os.path.join(HERE, src)], stdout=subprocess.DEVNULL)
        b = bytearray(open(out, "rb").read())
        b[34:38] = b"\0\0\0\0"
        open(out, "wb").write(bytes(b))
        print("wrote", out, len(b), "bytes")


if __name__ == "__main__":
    main()
