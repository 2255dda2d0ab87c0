"""Fingerprint of the kernel sources, so that measurements kept in files can be tied to the code they were taken on."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def csrc_hash():
    """sha256 (first 16 hex digits) over the names and contents of everything under csrc/ that is compiled into the
    library: the .hip unit, the headers it includes and the Makefile (flags change the code too).  Works from a bare
    copy of the tree - no git needed."""
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]
