"""Quick per-step timing of the PWC plan (dev tool; bench.py is the contract)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pwc_oracle as O
from premvos_amd.flow import pwc_dc_net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W = 512, 896
net = pwc_dc_net(None, use_graph=True); net.load_state_dict(O.synth_state_dict(0))
x = O.synth_frame_pair(H, W).repeat(B, 1, 1, 1).cuda()
for _ in range(3): net(x)
torch.cuda.synchronize()
t = time.time(); N = 20
for _ in range(N): net(x)
torch.cuda.synchronize()
dt = (time.time() - t) / N
print(f"graph forward B={B}: {dt*1e3:.3f} ms/iter  {B/dt:.1f} pairs/s  {168.2*B/dt/1e3:.1f} TFLOP/s")
p = net.plan(B, H, W)
# per-step timing
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in p.steps]
for rep in range(2):
    for (a, b), (name, fn) in zip(ev, p.steps):
        a.record(); fn(); b.record()
torch.cuda.synchronize()
tot = 0
rows = []
for (a, b), (name, fn) in zip(ev, p.steps):
    ms = a.elapsed_time(b); tot += ms; rows.append((ms, name))
print(f"sum of steps (eager, event-timed): {tot:.3f} ms over {len(p.steps)} launches")
for ms, name in sorted(rows, reverse=True)[:25]:
    print(f"  {ms*1e3:9.1f} us  {name}")
