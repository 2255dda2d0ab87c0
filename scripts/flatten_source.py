"""Prints the library source as one file: ascii_rast.hip with its local #include "part.h" lines replaced by
the parts (diagnostic variants are made by patching this text; see build_variant.sh, make_stamps_variant.py)."""
import os, re, sys
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ascii-renderer_amd", "csrc")


def flatten():
    out = []
    for line in open(os.path.join(CSRC, "ascii_rast.hip")).read().split("\n"):
        m = re.match(r'#include "(\w+\.h)"$', line)
        if m and os.path.exists(os.path.join(CSRC, m.group(1))):
            out.append(open(os.path.join(CSRC, m.group(1))).read().replace("#pragma once\n", ""))
        else:
            out.append(line)
    return "\n".join(out)


if __name__ == "__main__":
    sys.stdout.write(flatten())
