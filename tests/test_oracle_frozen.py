"""The oracle's outputs on committed inputs against committed digests (tests/golden/oracle_frozen*.{npz,json}, written by
tools/freeze_oracle.py).  The oracle is the working pin of the HIP path for every stage the reference holds no vectors for
(K1-K9, K11, the rank-3 pieces: DESIGN 2); this test is what makes an edit of the oracle that changes an answer visible --
the fixture has to change with it.  CPU only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import freeze_oracle as fz  # noqa: E402


def test_oracle_outputs_equal_the_frozen_digests():
    inputs = dict(np.load(fz.INPUTS))
    want = json.load(open(fz.DIGESTS))
    got = json.loads(json.dumps(fz.compute(inputs)))
    # the stored inputs themselves first (a corrupted fixture or a changed weight stream must not read as an oracle change)
    assert got["_inputs"] == want["_inputs"], "the committed input arrays do not match their digests"
    for name in ("cnn_random", "cnn_trained"):
        assert got[name]["weights"] == want[name]["weights"], "%s: the weights the digests were taken with are not reproduced" % name
    diff = fz.differences(want, got)
    assert not diff, "the oracle's answers changed -- if intended, rerun tools/freeze_oracle.py and commit the fixture:\n" + "\n".join(diff[:40])


def test_a_changed_answer_is_reported():
    """the comparison itself: one flipped digest, one float outside the tolerance, one inside it"""
    want = json.load(open(fz.DIGESTS))
    got = json.loads(json.dumps(want))
    scene = sorted(k for k in got if k.startswith("scene"))[0]
    got[scene]["median"] = "0" * 32 + got[scene]["median"][32:]
    got["cnn_random"]["goban3"]["pool2"]["values"][5] += 1e-3 * got["cnn_random"]["goban3"]["pool2"]["max"]
    got["cnn_random"]["goban3"]["pool4"]["values"][5] += 1e-8 * got["cnn_random"]["goban3"]["pool4"]["max"]
    diff = fz.differences(want, got)
    assert len(diff) == 2 and any("median" in d for d in diff) and any("pool2" in d for d in diff)
