#!/usr/bin/env python
"""Adds to premvos_amd/tune_gfx950.json the conv signatures of the FLOW net that the shipped table does not hold yet (a kernel
option that changes which filters a layer packs -- F(4x4,3x3) from 64 input channels, 32-channel outputs -- changes the layer's
signature): family and k-slices by the closed-form rule (ops.rule_choice), the order-neutral knobs timed once, here, instead of at
every plan build.  Existing entries are not touched.  Shapes: the ones tools/make_tune_table.py covers for the flow stage.

    python tools/extend_tune_table.py [--with-1080p]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("PREMVOS_TUNE_CACHE", None)
import torch  # noqa: E402

from premvos_amd import ops, synth  # noqa: E402
from premvos_amd.flow.driver import FlowStage  # noqa: E402


def main():
    before = {json.dumps(k) for k, _ in json.load(open(ops.TUNE_TABLE))}
    sd = synth.pwc_state_dict(0)
    shapes = [(480, 854, b) for b in (16, 8, 1, 2, 4)] + ([(1080, 1920, 16)] if "--with-1080p" in sys.argv else [])
    for h, w, b in shapes:
        fa, fb = synth.video_frames(b, h, w, 0)
        st = FlowStage(sd, batch=b, use_graph=False)
        st.run(fa.cuda(), fb.cuda())
        torch.cuda.synchronize()
        del st
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        print(f"flow stage {h}x{w} x {b}: {len(ops._TUNE_CACHE)} signatures", flush=True)
    ops.save_tune_cache(ops.TUNE_TABLE)
    table = json.load(open(ops.TUNE_TABLE))
    new = [(k, v) for k, v in table if json.dumps(k) not in before]
    for k, v in new:
        print("  +", k[:13], v)
    print(f"wrote {ops.TUNE_TABLE}: {len(table)} signatures ({len(new)} new)")


if __name__ == "__main__":
    main()
