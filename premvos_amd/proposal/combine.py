"""proposal_net/combine_general_and_specific.py:1-41: per frame, general proposals followed by specific ones.
A missing / unreadable file counts as an empty list (the reference's bare ``except``).

Frames that exist only in the specific set: the reference appends them to its work list, but its two ``replace(one_dir,
...)`` calls are no-ops on such a path, so nothing is written to combined_proposals/ for them (it rewrites the specific
file in place with its own content twice).  This module produces the same combined_proposals/ tree -- i.e. nothing for
those frames -- and leaves the input untouched (``tests/golden/host_refs.json`` holds the reference's output for exactly
this case)."""
from __future__ import annotations

import glob
import json
import os
from typing import List


def _load(fn: str) -> List[dict]:
    try:
        with open(fn, "r") as f:
            return json.load(f)
    except Exception:
        return []


def combine(root_dir: str = "./output/intermediate/", one_dir: str = "general_proposals/",
            two_dir: str = "specific_proposals/", out_dir: str = "combined_proposals/") -> int:
    files = sorted(glob.glob(os.path.join(root_dir, one_dir, "*", "*.json")))
    for f1 in files:
        fin = _load(f1) + _load(f1.replace(one_dir, two_dir))
        out_file = f1.replace(one_dir, out_dir)
        os.makedirs(os.path.dirname(out_file), exist_ok=True)
        with open(out_file, "w") as f:
            f.write(json.dumps(fin))
    return len(files)


if __name__ == "__main__":
    combine()
