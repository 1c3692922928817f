#!/usr/bin/env python
"""Where a rank's cold start goes (18.3 s): cProfile over what bench.py does between process start and the end
of its first launch -- synthetic weights, FramePipeline construction (packing, plans, arena, table lookup), first step.

    python tools/dev/cold_start_profile.py [batch] > gpurun_out/r06/cold_start_profile.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
t00 = time.perf_counter()
import torch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
marks = [("import torch", time.perf_counter() - t00)]


def main():
    t = time.perf_counter()
    from premvos_amd import synth
    from premvos_amd.pipeline import FramePipeline
    marks.append(("import premvos_amd", time.perf_counter() - t))
    t = time.perf_counter()
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    marks.append(("first CUDA call (context)", time.perf_counter() - t))
    t = time.perf_counter()
    w = (synth.pwc_state_dict(0), synth.proposal_weights(0), synth.proposal_weights(1), synth.refinement_weights(0))
    marks.append(("synthetic weights (bench only; a product run reads checkpoints)", time.perf_counter() - t))
    t = time.perf_counter()
    pipe = FramePipeline(*w, batch=B, device="cuda", boxes_per_frame=20)
    torch.cuda.synchronize()
    marks.append(("FramePipeline(...)", time.perf_counter() - t))
    clip = synth.clip_frames(0, B + 1, 480, 854).cuda()
    bx = synth.clip_boxes(0, B, 20, 480, 854).cuda()
    t = time.perf_counter()
    pipe.step(clip[:B].contiguous(), clip[1:B + 1].contiguous(), bx)
    torch.cuda.synchronize()
    marks.append(("first step (plans, arenas, table lookup, first launches)", time.perf_counter() - t))
    t = time.perf_counter()
    pipe.step(clip[:B].contiguous(), clip[1:B + 1].contiguous(), bx)
    torch.cuda.synchronize()
    marks.append(("second step", time.perf_counter() - t))


pr = cProfile.Profile()
pr.enable()
main()
pr.disable()
for k, v in marks:
    print(f"{v:8.2f} s  {k}")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
print(s.getvalue()[:14000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:9000])
