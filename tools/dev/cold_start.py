"""Dev tool (round 5): where a rank's cold start goes (bench.py `cold_start_s`): weights, packing, plans + graphs, first launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t0 = time.perf_counter()
import torch
from premvos_amd import synth
from premvos_amd.pipeline import FramePipeline
t = [("import torch + package", time.perf_counter() - t0)]


def mark(what, t1):
    torch.cuda.synchronize()
    t.append((what, time.perf_counter() - t1))
    return time.perf_counter()


t1 = time.perf_counter()
torch.zeros(1, device="cuda")
t1 = mark("HIP context", t1)
sd, pg, ps, rw = synth.pwc_state_dict(0), synth.proposal_weights(0), synth.proposal_weights(1), synth.refinement_weights(0)
t1 = mark("synthetic weights on the host (a real job reads checkpoints instead)", t1)
pipe = FramePipeline(sd, pg, ps, rw, batch=16, device="cuda", boxes_per_frame=20)
t1 = mark("FramePipeline(): weight packing (BN folding, Winograd filter transforms in fp64), upload", t1)
fa = synth.clip_frames(0, 17, 480, 854).cuda()
bx = synth.clip_boxes(0, 16, 20, 480, 854).cuda()
t1 = mark("synthetic frames", t1)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
pipe.step(fa[:16].contiguous(), fa[1:17].contiguous(), bx)
pr.disable()
t1 = mark("first step: plans (two-pass arena), table lookup, graph capture, first launch", t1)
pipe.step(fa[:16].contiguous(), fa[1:17].contiguous(), bx)
t1 = mark("second step", t1)
for w, s in t:
    print(f"{s:8.2f} s  {w}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
