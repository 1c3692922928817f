"""Dev tool: the pure-MFMA calibration kernel alone, N launches (for PMC clock readings)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import _lib
lib, st = _lib.load(), _lib.current_stream()
sink = torch.zeros(4, device="cuda")
iters = 4000
for blocks in (2048,):
    for _ in range(3):
        _lib.check(lib.premvos_mfma_f32_calibrate(iters, blocks, sink.data_ptr(), st))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        lib.premvos_mfma_f32_calibrate(iters, blocks, sink.data_ptr(), st)
    b.record(); b.synchronize()
    us = a.elapsed_time(b) * 100
    print(f"calib blocks={blocks}: {us:.1f} us {blocks * 4 * iters * 16 * 4096 / us / 1e6:.1f} TF/s")
