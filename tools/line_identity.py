#!/usr/bin/env python3
"""Line-level identity of the host mirrors against their namesakes in the reference tree (dev
tool, run in the build container only: /root/reference does not travel).  The measure is the one
the round-1 review used: non-blank, non-comment lines longer than 12 characters that also occur
(after whitespace normalisation) in the reference file, over the file's own such lines."""
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/camkifu"
PAIRS = [
    ("camkifu_amd/core/video.py", "core/video.py"),
    ("camkifu_amd/core/vmanager.py", "core/vmanager.py"),
    ("camkifu_amd/core/capture.py", "core/vmanager.py"),
    ("camkifu_amd/board/boardfinder.py", "board/boardfinder.py"),
    ("camkifu_amd/board/bf_auto.py", "board/bf_auto.py"),
    ("camkifu_amd/stone/stonesfinder.py", "stone/stonesfinder.py"),
    ("camkifu_amd/stone/sf_neural.py", "stone/sf_neural.py"),
    ("camkifu_amd/stone/sf_contours.py", "stone/sf_contours.py"),
    ("camkifu_amd/stone/nn_manager.py", "stone/nn_manager.py"),
    ("camkifu_amd/stone/nn_cache.py", "stone/nn_cache.py"),
    ("camkifu_amd/stone/policy.py", "stone/sf_neural.py"),
    ("camkifu_amd/stone/gridfit.py", "stone/stonesfinder.py"),
    ("camkifu_amd/pipeline.py", "stone/sf_neural.py"),
    ("oracle/ora_logic.py", "stone/sf_neural.py"),
    ("oracle/ora_logic.py", "core/imgutil.py"),
    ("oracle/ora_stones.py", "stone/sf_contours.py"),
    ("oracle/ora_grid.py", "stone/stonesfinder.py"),
    ("oracle/ora_grid.py", "core/imgutil.py"),
]


def lines_of(path):
    out = []
    with open(path, encoding="utf-8", errors="replace") as fh:
        in_doc = False
        for raw in fh:
            s = re.sub(r"\s+", " ", raw.strip())
            if s.count('"""') == 1 or s.count("'''") == 1:
                in_doc = not in_doc
                continue
            if in_doc or not s or s.startswith("#") or s.startswith('"""') or s.startswith("'''"):
                continue
            s = re.sub(r"\s*#.*$", "", s)
            if len(s) > 12:
                out.append(s)
    return out


def main():
    worst = 0.0
    for mine, ref in PAIRS:
        a, b = os.path.join(REPO, mine), os.path.join(REF, ref)
        if not (os.path.exists(a) and os.path.exists(b)):
            continue
        la, lb = lines_of(a), set(lines_of(b))
        same = [s for s in la if s in lb]
        pct = 100.0 * len(same) / max(1, len(la))
        worst = max(worst, pct)
        print("%-40s %3d/%3d  %5.1f %%" % (mine, len(same), len(la), pct))
        if "-v" in sys.argv:
            for s in same:
                print("      | " + s)
    return 0 if worst < 15.0 else 1


if __name__ == "__main__":
    sys.exit(main())
