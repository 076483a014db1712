"""Loading of tests/golden/*.json (hex -> numpy uint8)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def hb(h):
    return np.frombuffer(bytes.fromhex(h), dtype=np.uint8).copy()


def csr(d):
    return (np.array(d["row_ptr"], dtype=np.uint64), np.array(d["col"], dtype=np.uint32), hb(d["val_mont"]))
