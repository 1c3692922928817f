"""Dev tool: device memory of the bench's pipeline object stage by stage (torch allocator, GiB): python tools/dev/mem_report.py [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import synth
from premvos_amd.flow.driver import FlowStage
from premvos_amd.proposal.driver import ProposalStage
from premvos_amd.refinement.model import RefinementNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W, P = 480, 854, 20
g = lambda: torch.cuda.memory_allocated() / 2 ** 30
last = [0.0]


def mark(what):
    torch.cuda.synchronize()
    now = g()
    print(f"{what:60s} +{now - last[0]:7.2f} GiB   total {now:7.2f}   peak {torch.cuda.max_memory_allocated() / 2 ** 30:7.2f}", flush=True)
    last[0] = now


fa = synth.clip_frames(0, B, H, W).cuda()
fb = synth.clip_frames(1, B + 1, H, W).cuda()
boxes = synth.clip_boxes(0, B, P, H, W).cuda()
mark("inputs")
flow = FlowStage(synth.pwc_state_dict(0), batch=B, device="cuda")
flow.run(fa, fb)
mark("flow stage (plan of 16 pairs at 512x896, Winograd workspace, kept slab)")
pg = ProposalStage(synth.proposal_weights(0), batch=B, device="cuda", rgb_input=True)
pg.run(fa)
mark("proposal stage, one weight set (749x1333, 100 RoIs per frame)")
ref = RefinementNet(synth.refinement_weights(0), 16, "cuda")
mark("refinement net weights")
p = ref.refine_group(fa[:8], boxes[:8], lane=0)
mark("refinement plan: 8 frames x 20 boxes, lane 0")
p = ref.refine_group(fa[:8], boxes[:8], lane=1)
mark("refinement plan: same, lane 1")
p = ref.refine_group(fa[:4], boxes[:4], lane=0)
mark("refinement plan: 4 frames x 20 boxes, lane 0")
