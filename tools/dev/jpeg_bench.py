"""Dev tool: the optional GPU JPEG decode against PIL on the bench's synthetic 480p / 1080p frames (one host thread)."""
import io, os, sys, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import jpeg, synth

for h, w in ((480, 854), (1080, 1920)):
    fr = synth.video_frames(1, h, w + (-w) % 8)[0][0].numpy()[:, :w]
    b = io.BytesIO(); Image.fromarray(fr).save(b, "JPEG", quality=95); data = b.getvalue()
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    assert np.array_equal(jpeg.decode(data).cpu().numpy(), ref)
    n = 50
    t = time.time()
    for _ in range(n): np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    t_pil = (time.time() - t) / n
    t = time.time()
    for _ in range(n): torch.from_numpy(np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))).cuda()
    torch.cuda.synchronize(); t_pil_up = (time.time() - t) / n
    t = time.time()
    ds = [jpeg.entropy_decode(data) for _ in range(n)]
    t_host = (time.time() - t) / n
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.time(); a.record()
    outs = [jpeg.reconstruct(d) for d in ds]
    b_.record(); torch.cuda.synchronize(); t_dev_wall = (time.time() - t) / n
    print(f"{h}x{w} ({len(data) / 1e3:.0f} KB): PIL decode {t_pil * 1e3:.2f} ms, PIL + upload {t_pil_up * 1e3:.2f} ms | host Huffman {t_host * 1e3:.2f} ms, "
          f"GPU half {a.elapsed_time(b_) / n * 1e3:.0f} us GPU time / {t_dev_wall * 1e3:.2f} ms host wall per frame")
