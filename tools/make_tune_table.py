#!/usr/bin/env python
"""Regenerates premvos_amd/tune_gfx950.json -- the conv configuration table that ships with the package -- on an MI355X:
every conv signature of the shapes the product runs is explored by wall clock ONCE, here (PREMVOS_AUTOTUNE=full), and the
winners are frozen into the file; at run time nothing that changes the order of an fp32 sum is ever decided by a stopwatch
again (premvos_amd.ops.autotune).  Shapes covered (480x854 frames, full-depth nets):

  bench.py                      FramePipeline, 16 frames per step, refinement groups of 8 frames x 20 boxes
  premvos_amd.stream            chunks of 8 frames (flow / proposals), packed refinement groups of <= 4 frames (80 ... 112 slots)
  the stage drivers             1 frame per launch list (flow, proposals), refinement of 1 frame (18 ... 26 boxes) and packed groups
  bench.py --frame 1080p        (with --with-1080p)

    python tools/make_tune_table.py [--out premvos_amd/tune_gfx950.json] [--with-1080p] [--quick] [--polish]

A full exploration times each candidate 2 x 4 times: its ~1 % noise decides between near-equal configurations, so re-running it
after a kernel change can LOSE (round 3: -2 % on the pipeline in a same-box A/B).  --polish is the safe refresh.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
POLISH = "--polish" in sys.argv       # keep every entry's arithmetic, re-time only its order-neutral knobs (ops._polish)
os.environ["PREMVOS_AUTOTUNE"] = "polish" if POLISH else "full"
if not POLISH:
    os.environ["PREMVOS_TUNE_TABLE"] = "0"
os.environ.pop("PREMVOS_TUNE_CACHE", None)
if "--reps" in sys.argv:               # launches per timing burst of a full exploration (default 4: ~1 % noise between near-equal
    os.environ["PREMVOS_AUTOTUNE_REPS"] = sys.argv[sys.argv.index("--reps") + 1]      # configurations; 16 takes ~4x as long)

import torch  # noqa: E402

from premvos_amd import ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=ops.TUNE_TABLE)
    ap.add_argument("--with-1080p", action="store_true")
    ap.add_argument("--quick", action="store_true", help="bench shapes only")
    ap.add_argument("--reps", type=int, default=4, help="launches per timing burst of a full exploration")
    ap.add_argument("--polish", action="store_true", help="start from the shipped table; re-time only order-neutral knobs (12 x 3 "
                    "repetitions), replace an entry when another block / stage depth is >= 1.5 %% faster; results do not change")
    a = ap.parse_args()
    from premvos_amd.flow.driver import FlowStage
    from premvos_amd.pipeline import FramePipeline
    from premvos_amd.proposal.driver import ProposalStage
    from premvos_amd.refinement.model import RefinementNet
    t0 = time.time()
    H, W = 480, 854

    def free():
        import gc
        gc.collect()                     # plans hold their buffers through closures (reference cycles)
        torch.cuda.empty_cache()

    def note(what):
        torch.cuda.synchronize()
        free()
        print(f"[{time.time() - t0:6.0f} s] {what}: {len(ops._TUNE_CACHE)} signatures, "
              f"{torch.cuda.memory_allocated() / 2**30:.1f} GiB held", flush=True)

    sd, pg, ps, rw = synth.pwc_state_dict(0), synth.proposal_weights(0), synth.proposal_weights(1), synth.refinement_weights(0)
    for (h, w) in ([(H, W)] + ([(1080, 1920)] if a.with_1080p else [])):
        pipe = FramePipeline(sd, pg, ps, rw, batch=16, boxes_per_frame=20)
        fa, fb = synth.video_frames(16, h, w, 0)
        pipe.step(fa.cuda(), fb.cuda(), synth.boxes(16, 20, h, w, 0).cuda())
        del pipe
        note(f"bench pipeline {h}x{w}")
    if not a.quick:
        fa, fb = synth.video_frames(8, H, W, 0)
        fa, fb = fa.cuda(), fb.cuda()
        for b in (8, 1, 2, 4):
            st = FlowStage(sd, batch=b, use_graph=False)
            st.run(fa[:b], fb[:b])
            del st
            free()
        note("flow stages")
        for b in (8, 1, 2, 4):
            st = ProposalStage(pg, batch=b, rgb_input=True, use_graph=False)
            st.run(fa[:b])
            del st
            free()
        note("proposal stages")
        net = RefinementNet(rw, 16, use_graph=False)
        boxes = synth.boxes(4, 40, H, W, 3)
        for slots in (80, 88, 96, 104, 112, 64, 72, 48, 56, 120):
            per = [boxes[g, :slots // 4] for g in range(4)]
            net.refine_packed(fa[:4], per, slots, 4)
            net._plans.clear()
            free()
        note("packed refinement groups")
        for p in (20, 22, 24, 26, 18, 16, 28, 32, 40):
            net.refine(fa[0], boxes[0, :p].cuda(), max_boxes=p)
            net._plans.clear()
            free()
        note("single-frame refinement")
    ops.save_tune_cache(a.out)
    table = json.load(open(a.out))
    fam = {}
    for _, v in table:
        k = {1: "direct", 2: "wino2x2_slab", 3: "wino2x2_fused", 4: "wino4x4"}.get(v[0], "igemm")
        fam[k] = fam.get(k, 0) + 1
    print(f"wrote {a.out}: {len(table)} signatures, {fam}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
