#!/usr/bin/env python3
"""Hash of the sources libphyloligo_amd.so is built from: csrc/*.hip, csrc/*.h, csrc/*.cpp, csrc/Makefile and
include/phyloligo_amd.h (names and contents, sorted; first 16 hex digits of the SHA-256).  The Makefile bakes it into
po_version(); tools/pmc_busy.py / pmc_traffic.py write the hash of the library they profiled into profiles/*.json; bench.py
and the tests compare the two, so that counters measured on another build are flagged.
    source_hash.py              print the hash
    source_hash.py --header F   (re)write F with `#define PO_SRC_HASH "<hash>"` if it differs (used by the Makefile)"""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files(root=ROOT):
    csrc = os.path.join(root, "phyloligo_amd", "csrc")
    files = [f for pat in ("*.hip", "*.h", "*.cpp") for f in glob.glob(os.path.join(csrc, pat))]
    files += [os.path.join(csrc, "Makefile"), os.path.join(root, "include", "phyloligo_amd.h")]
    return sorted(files)


def source_hash(root=ROOT):
    h = hashlib.sha256()
    for f in source_files(root):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    digest = source_hash()
    if len(sys.argv) == 3 and sys.argv[1] == "--header":
        text = '#define PO_SRC_HASH "%s"\n' % digest
        try:
            same = open(sys.argv[2]).read() == text
        except OSError:
            same = False
        if not same:
            os.makedirs(os.path.dirname(sys.argv[2]), exist_ok=True)
            with open(sys.argv[2], "w") as fh:
                fh.write(text)
    else:
        print(digest)
