#!/usr/bin/env python
"""SURVEY 8(f3) acceptance, one command on a box that HAS the released weights and DAVIS (neither is in the build image):

    python tools/accept_davis.py --root <PReMVOS root> [--gpus N] [--reference-python python2.7-or-3-with-the-reference-env]

  1. checks that the inputs exist where simple_run.sh expects them (weights/PReMVOS_weights/..., data/DAVIS/JPEGImages/480p,
     seq_to_run.txt) and that every checkpoint can be READ and holds the variables the nets ask for (premvos_amd.weights, checksums
     verified) -- without running anything;
  2. runs this package's stages on them: python -m premvos_amd.stream (flow, proposals x2, combine, refinement; --gpus N) and the
     ReID stage (premvos_amd.reid.driver) -> output/intermediate/{flow,*_proposals}/;
  3. MergeTrack stays the reference's (out of scope here): runs `code/MergeTrack/merge.py` with --reference-python
     when given, otherwise expects output/final/ to exist already or stops with the command to run;
  4. evaluates output/final/<seq>/*.png against data/DAVIS/Annotations/480p with tools/davis_eval.py and compares with the
     reference's README.md:35-38 numbers (J 0.7363, F 0.80044, J&F 76.8366) within --tolerance.

Exit code 0 = within tolerance, 1 = outside, 2 = an input is missing (the message names it)."""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

README_NUMBERS = {"mean_J": 0.7363, "mean_F": 0.80044, "mean_JF_percent": 76.8366}      # /root/reference/README.md:35-38
WEIGHTS = {
    "flow": ("weights/PReMVOS_weights/optical_flow_net/pwc_net.pth.tar", None),
    "general": ("weights/PReMVOS_weights/proposal_net/general_weights/proposal_general_weights", "proposal"),
    "specific": ("weights/PReMVOS_weights/proposal_net/specific_weights/proposal_specific_weights", "proposal"),
    "refinement": ("weights/PReMVOS_weights/refinement_net/specific_weights/refinement_specific_weights", "refinement"),
    "reid": ("weights/PReMVOS_weights/ReID_net/ReID_general_weights", "reid"),
}


def check_inputs(root: str, skip_reid: bool) -> list:
    """-> list of problems (empty = ready).  Reads every checkpoint through the product's own loaders."""
    from premvos_amd import weights as W
    problems = []
    if not os.path.exists(os.path.join(root, "seq_to_run.txt")):
        problems.append("seq_to_run.txt is missing (simple_run.sh:21-26 reads the sequence folders from it)")
    else:
        for ln in open(os.path.join(root, "seq_to_run.txt")):
            d = os.path.join(root, ln.strip())
            if ln.strip() and not os.path.isdir(d):
                problems.append(f"sequence folder {d} (from seq_to_run.txt) does not exist")
    for tag, (rel, kind) in WEIGHTS.items():
        path = os.path.join(root, rel)
        if tag == "reid" and skip_reid:
            continue
        try:
            if kind is None:
                import torch
                sd = torch.load(path, map_location="cpu")
                sd = sd.get("state_dict", sd)
                if "conv1a.0.weight" not in sd and "module.conv1a.0.weight" not in sd:
                    problems.append(f"{path}: no PWC-Net state_dict inside (conv1a.0.weight missing)")
            elif kind == "reid":
                v = W.load_tf_checkpoint(path)
                if not W.reid_weights_from_tf(v):
                    problems.append(f"{path}: no ReID variables found")
            else:
                W.load_any(path, kind)
        except Exception as e:                      # noqa: BLE001 -- every reader error becomes one line of the report
            problems.append(f"{tag}: {type(e).__name__}: {e}")
    return problems


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--root", default=".")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--reference-python", default=None, help="interpreter that can run the reference's code/MergeTrack/merge.py")
    ap.add_argument("--annotations", default="data/DAVIS/Annotations/480p")
    ap.add_argument("--tolerance", type=float, default=0.5, help="allowed |J&F - 76.8366| in percent points")
    ap.add_argument("--skip-stages", action="store_true", help="only evaluate an existing output/final/")
    ap.add_argument("--skip-reid", action="store_true")
    ap.add_argument("--check-only", action="store_true", help="stop after step 1")
    a = ap.parse_args(argv)
    root = os.path.abspath(a.root)
    if not a.skip_stages:
        problems = check_inputs(root, a.skip_reid)
        if problems:
            print("accept_davis: inputs are not ready:\n  " + "\n  ".join(problems))
            return 2
        print("accept_davis: all checkpoints read, every variable the nets ask for is present")
        if a.check_only:
            return 0
        env = dict(os.environ, PYTHONPATH=os.path.dirname(HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
        subprocess.check_call([sys.executable, "-m", "premvos_amd.stream", "--root", root, "--gpus", str(a.gpus)], env=env)
        if not a.skip_reid and not os.path.isdir(os.path.join(root, "output/intermediate/ReID_proposals")):
            from premvos_amd.reid import driver as qd
            cwd = os.getcwd()
            os.chdir(os.path.join(root, "code"))
            try:
                qd.main(["ReID_net/configs/run"])
            finally:
                os.chdir(cwd)
    final = os.path.join(root, "output", "final")
    if not os.path.isdir(final):
        if a.reference_python:
            subprocess.check_call([a.reference_python, "MergeTrack/merge.py"], cwd=os.path.join(root, "code"))
        else:
            print("accept_davis: output/final/ does not exist.  MergeTrack is the reference's own program (out of scope here);\n"
                  f"  run it on the intermediate results:   cd {root}/code && ./MergeTrack/merge.py\n"
                  "  (or pass --reference-python <interpreter with its requirements>), then call this script with --skip-stages")
            return 2
    ann = os.path.join(root, a.annotations)
    if not os.path.isdir(ann):
        print(f"accept_davis: annotations {ann} are missing")
        return 2
    import davis_eval
    seqs = sorted(d for d in os.listdir(final) if os.path.isdir(os.path.join(final, d)))
    r = davis_eval.evaluate(final, ann, seqs)
    r["reference_readme"] = README_NUMBERS
    r["delta_JF_percent"] = round(r["mean_JF_percent"] - README_NUMBERS["mean_JF_percent"], 4)
    print(json.dumps({k: v for k, v in r.items() if k != "per_sequence"}, indent=1))
    with open(os.path.join(root, "output", "premvos_amd_davis_eval.json"), "w") as f:
        json.dump(r, f, indent=1)
    return 0 if abs(r["delta_JF_percent"]) <= a.tolerance else 1


if __name__ == "__main__":
    sys.exit(main())
