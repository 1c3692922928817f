#!/usr/bin/env python
"""DEV HARNESS (not product code; used by tests/test_gpu_plumbing.py and tools/time_drivers.py): chains the stage drivers
like simple_run.sh:21-68 for the stages on the hot path (A flow, B proposals x2, C combine, D refinement, + the ReID batch
stage), with the script's own directory-exists resume.  MergeTrack (simple_run.sh:70-77) is out of scope and keeps running
from the reference on these outputs.  Original docstring: counterpart of simple_run.sh:21-58 (`if [ ! -d ... ]`).  Run from the PReMVOS root: same relative inputs
(seq_to_run.txt, data/DAVIS/JPEGImages/480p/<seq>/) and outputs (output/intermediate/{flow,general_proposals,
specific_proposals,combined_proposals,refined_proposals}/<seq>/...) that the unchanged ReID and MergeTrack stages read.

Weights: the reference's own files at the reference's own paths -- optical_flow_net/pwc_net.pth.tar (torch pickle) and the
TF checkpoint prefixes of the two TF nets (simple_run.sh:31-32,39-40; refinement_net/configs/run:9), read without
TensorFlow by premvos_amd/weights.py; torch pickles of name->tensor dicts are accepted too.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Optional

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=".")
    ap.add_argument("--seq_file", default="seq_to_run.txt")
    ap.add_argument("--flow_weights", default="weights/PReMVOS_weights/optical_flow_net/pwc_net.pth.tar")
    ap.add_argument("--general_weights", default="weights/PReMVOS_weights/proposal_net/general_weights/proposal_general_weights")
    ap.add_argument("--specific_weights", default="weights/PReMVOS_weights/proposal_net/specific_weights/proposal_specific_weights")
    ap.add_argument("--refinement_weights", default="weights/PReMVOS_weights/refinement_net/specific_weights/refinement_specific_weights")
    ap.add_argument("--reid_config", default="code/ReID_net/configs/run",
                    help="ReID config (network table + 'load'); its relative paths are taken from the config's grandparent "
                         "directory, like the reference which runs the stage from code/")
    a = ap.parse_args(argv)
    os.chdir(a.root)
    inter = "output/intermediate"
    done = []

    flow_loc = f"{inter}/flow"
    if not os.path.isdir(flow_loc):                                   # simple_run.sh:22-26
        from premvos_amd.flow import driver as fd
        fd.main([a.seq_file, a.flow_weights, flow_loc])
        done.append("flow")
    from premvos_amd.proposal import driver as pd
    for loc, wfile in ((f"{inter}/general_proposals", a.general_weights),          # :28-42
                       (f"{inter}/specific_proposals", a.specific_weights)):
        if not os.path.isdir(loc):
            pd.main(["--forward", loc, "--agnostic", "--second_head", "--forward_dataset", "DAVIS", "--load", wfile,
                     "--davis_name", os.path.join(os.getcwd(), a.seq_file)])
            done.append(os.path.basename(loc))
    comb = f"{inter}/combined_proposals"
    if not os.path.isdir(comb):                                       # :44-49
        from premvos_amd.proposal.combine import combine
        combine(inter + "/")
        done.append("combined_proposals")
    refined = f"{inter}/refined_proposals"
    if not os.path.isdir(refined):                                    # :51-58
        from premvos_amd.refinement import driver as rd
        w = rd.load_weights(a.refinement_weights)
        eng = rd.RefinementEngine(rd.RefinementNet(w, rd.infer_num_middle(w)))
        rd.forward_directory(eng, "data/DAVIS/JPEGImages/480p/", comb + "/", refined + "/")
        done.append("refined_proposals")
    reid = f"{inter}/ReID_proposals"
    if not os.path.isdir(reid) and os.path.exists(a.reid_config):     # :60-68
        from premvos_amd.reid import driver as qd
        cfg = qd.Config(a.reid_config)
        base = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(a.reid_config))))     # .../code
        if not os.path.isabs(cfg.str("load")):
            cfg._entries["load"] = os.path.normpath(os.path.join(base, cfg.str("load")))
        eng = qd.engine_from_config(cfg)
        qd.forward_directory(eng, "data/DAVIS/JPEGImages/480p/", refined + "/", reid + "/")
        done.append("ReID_proposals")
    print("stages run:", done)
    return 0


if __name__ == "__main__":
    sys.exit(main())
