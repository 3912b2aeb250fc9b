"""The golden-vector cases of tests/golden/ (inputs regenerate from seeds)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import CASES  # noqa: E402,F401

from adflow_amd.params import FlowParams  # noqa: E402
from adflow_amd.synth import make_block  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    dims, pk, mk, turb = CASES[name]
    prm = FlowParams(**pk)
    blk = make_block(*dims, prm, **mk)
    gold = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return prm, blk, gold, turb
