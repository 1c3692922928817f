"""proposal_net/combine_general_and_specific.py:1-41: per frame, general proposals followed by specific ones.
A missing / unreadable file counts as an empty list (the reference's bare ``except``); frames that exist only in the
specific set are added too."""
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
    translated = {f.replace(one_dir, two_dir) for f in files}
    files += [f.replace(two_dir, one_dir) for f in sorted(glob.glob(os.path.join(root_dir, two_dir, "*", "*.json")))
              if f not in translated]
    for f1 in files:
        fin = _load(f1) + _load(f1.replace(one_dir, two_dir))
        out_file = f1.replace(one_dir, out_dir)
        os.makedirs(os.path.dirname(out_file), exist_ok=True)
        with open(out_file, "w") as f:
            json.dump(fin, f)
    return len(files)


if __name__ == "__main__":
    combine()
