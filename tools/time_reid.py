"""Dev tool: timing of the full-depth ReID net, P boxes of one 480x854 frame."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import reid_oracle as R
from premvos_amd.reid import ReIDNet
P = int(sys.argv[1]) if len(sys.argv) > 1 else 40
net = ReIDNet(R.synth_weights(0))
rng = np.random.default_rng(0)
frame = torch.from_numpy(rng.integers(0, 256, (480, 854, 3), dtype=np.uint8)).cuda()
wh = rng.uniform(40, 400, (P, 2)); xy = rng.uniform(0, 1, (P, 2)) * (np.array([854, 480]) - np.minimum(wh, [854, 480]))
boxes = np.concatenate([xy, np.minimum(wh, [854, 480])], 1).astype(np.float32)
for _ in range(2): e = net.embed(frame, boxes, max_boxes=P)
torch.cuda.synchronize(); t = time.time(); N = 5
for _ in range(N): e = net.embed(frame, boxes, max_boxes=P)
torch.cuda.synchronize(); dt = (time.time() - t) / N
p = net.plan(P, 480, 854, True)
fl = sum(p.flops.values())
print(f"ReID P={P}: {dt*1e3:.2f} ms/frame  conv {fl/1e9:.1f} GFLOP -> {fl/dt/1e12:.1f} TFLOP/s; |emb| {float(e.abs().mean()):.3f}")
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in p.steps]
for (a, b), (n, f) in zip(ev, p.steps):
    a.record(); f(); b.record()
torch.cuda.synchronize()
rows = [(a.elapsed_time(b), n) for (a, b), (n, f) in zip(ev, p.steps)]
tc = sum(ms for ms, n in rows if n.startswith('conv:'))
print(f"sum eager {sum(r[0] for r in rows):.2f} ms, {len(rows)} launches; conv {tc:.2f} ms ({fl/tc/1e9:.1f} TF/s)")
for ms, n in sorted(rows, reverse=True)[:12]:
    print(f"  {ms*1e3:9.1f} us  {n:40s} {p.flops[n]/ms/1e9:6.1f} TF/s" if n in p.flops else f"  {ms*1e3:9.1f} us  {n}")
