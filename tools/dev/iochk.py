import os, sys, time, tempfile, numpy as np
sys.path.insert(0, '.')
from PIL import Image
from premvos_amd import io_pipeline as iop
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
base = (rng.random((60, 107, 3)) * 255).astype(np.uint8)
img = np.asarray(Image.fromarray(base).resize((854, 480), Image.BILINEAR))
fns = []
for i in range(64):
    fn = f"{d}/{i:05d}.jpg"; Image.fromarray(np.roll(img, i, 1)).save(fn, quality=95); fns.append(fn)
dec = lambda fn: np.ascontiguousarray(np.asarray(Image.open(fn).convert("RGB")))
t = time.time(); [dec(f) for f in fns]; t1 = time.time() - t
t = time.time(); list(iop.prefetch(fns, dec)); t2 = time.time() - t
print(f"decode serial {64/t1:.0f}/s, prefetch pool ({iop.io_threads()} threads) {64/t2:.0f}/s, tmpdir {d}")
x = np.zeros((480, 854, 2), np.float32)
t = time.time()
for i in range(64):
    with open(f"{d}/{i}.flo", "wb") as f: f.write(b"PIEH"); f.write(x.tobytes())
print(f"flo write {64/(time.time()-t):.0f}/s")
import torch
a = torch.zeros((8, 480, 854, 2), device="cuda")
torch.cuda.synchronize(); t = time.time()
for _ in range(8): b = a.cpu()
print(f"D2H 8x3.3MB pageable: {8*8/(time.time()-t):.0f} frames/s")
h = torch.from_numpy(np.zeros((8, 480, 854, 3), np.uint8))
t = time.time()
for _ in range(8): g = h.cuda(); torch.cuda.synchronize()
print(f"H2D 8 frames u8: {64/(time.time()-t):.0f} frames/s")
