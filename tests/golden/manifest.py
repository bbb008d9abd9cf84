#!/usr/bin/env python
"""Content manifest of the golden fixtures: tests/golden/MANIFEST.sha256.

    python tests/golden/manifest.py            # rewrite the manifest from the .npz files on disk
    python tests/golden/manifest.py --check    # exit 1 if a fixture is missing, extra or changed

One line per fixture: `<sha256>  <file>`.  The hash is over the fixture's CONTENT — for every array in key order: name, dtype,
shape, C-order bytes — not over the .npz container, whose zip headers carry the time of writing: a regenerated fixture with the
same arrays keeps its line (the generators are deterministic), a fixture whose numbers moved does not.  Checked on CPU by
tests/test_oracle_golden.py::test_golden_manifest.
"""
import glob
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "MANIFEST.sha256")


def content_hash(path):
    h = hashlib.sha256()
    with np.load(path, allow_pickle=False) as z:
        for k in sorted(z.files):
            a = np.ascontiguousarray(z[k])
            h.update(f"{k}|{a.dtype.str}|{a.shape}|".encode())
            h.update(a.tobytes())
    return h.hexdigest()


def current():
    return {os.path.basename(f): content_hash(f) for f in sorted(glob.glob(os.path.join(HERE, "*.npz")))}


def read():
    out = {}
    for line in open(PATH):
        if line.strip() and not line.startswith("#"):
            h, name = line.split()
            out[name] = h
    return out


def write():
    cur = current()
    with open(PATH, "w") as f:
        f.write("# sha256 over array contents (name | dtype | shape | bytes, keys sorted), see tests/golden/manifest.py\n")
        for name, h in cur.items():
            f.write(f"{h}  {name}\n")
    print(f"wrote {PATH}: {len(cur)} fixtures")


def diff():
    """(missing, extra, changed) fixture names, manifest vs disk."""
    want, have = read(), current()
    return (sorted(set(want) - set(have)), sorted(set(have) - set(want)), sorted(k for k in want if k in have and want[k] != have[k]))


if __name__ == "__main__":
    if "--check" in sys.argv:
        d = diff()
        print("missing %s\nextra %s\nchanged %s" % d)
        sys.exit(1 if any(d) else 0)
    write()
