"""Dev tool: per-step timing of the full-depth refinement_net, P boxes on a 480x854 frame."""
import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refinement_oracle as R
from premvos_amd.refinement import RefinementNet
P = int(sys.argv[1]) if len(sys.argv) > 1 else 20
net = RefinementNet(R.synth_weights(0, 16), 16)
rng = np.random.default_rng(0)
frame = torch.from_numpy(rng.integers(0, 256, (480, 854, 3), dtype=np.uint8)).cuda()
wh = rng.uniform(40, 400, (P, 2)); xy = rng.uniform(0, 1, (P, 2)) * (np.array([854, 480]) - np.minimum(wh, [854, 480]))
boxes = torch.tensor(np.stack([xy[:, 1], xy[:, 0], np.minimum(xy[:, 1] + wh[:, 1], 480), np.minimum(xy[:, 0] + wh[:, 0], 854)], 1), dtype=torch.float32).cuda()
for _ in range(2): p = net.refine(frame, boxes, max_boxes=P)
torch.cuda.synchronize(); t = time.time(); N = 5
for _ in range(N): p = net.refine(frame, boxes, max_boxes=P)
torch.cuda.synchronize(); dt = (time.time() - t) / N
fl = sum(p.flops.values())
print(f"refine P={P}: {dt*1e3:.2f} ms/frame  conv {fl/1e9:.1f} GFLOP -> {fl/dt/1e12:.1f} TFLOP/s; mem {torch.cuda.memory_allocated()/2**30:.2f} GiB; conf {p.conf[:3].tolist()}")
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in p.steps]
for (a, b), (n, f) in zip(ev, p.steps):
    a.record(); f(); b.record()
torch.cuda.synchronize()
rows = [(a.elapsed_time(b), n) for (a, b), (n, f) in zip(ev, p.steps)]
tc = sum(ms for ms, n in rows if n.startswith('conv:')); td = sum(ms for ms, n in rows if n.startswith('dw:'))
db = sum(p.dw_bytes.values())
print(f"sum eager {sum(r[0] for r in rows):.2f} ms, {len(rows)} launches; conv {tc:.2f} ms ({fl/tc/1e9:.1f} TF/s); dw {td:.2f} ms ({db/td/1e6:.0f} GB/s algorithmic)")
for ms, n in sorted(rows, reverse=True)[:int(os.environ.get('TOP', '16'))]:
    extra = f"{p.flops[n]/ms/1e9:6.1f} TF/s" if n in p.flops else (f"{p.dw_bytes[n]/ms/1e6:6.0f} GB/s" if n in p.dw_bytes else "")
    print(f"  {ms*1e3:9.1f} us  {n:70s} {extra}")

if os.environ.get("NONCONV"):
    print("non-conv launches:")
    for ms, n in sorted([r for r in rows if not r[1].startswith("conv:")], reverse=True)[:40]:
        extra = f"{p.dw_bytes[n]/ms/1e6:6.0f} GB/s" if n in p.dw_bytes else ""
        print(f"  {ms*1e3:9.1f} us  {n:70s} {extra}")
