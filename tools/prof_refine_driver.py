"""Dev tool: cProfile of the refinement batch stage on a synthetic 480p sequence (where does the host time go?)."""
import cProfile, json, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from oracle import pwc_oracle as O, refinement_oracle as RO
from premvos_amd.refinement import driver as rd
T = 48
root = tempfile.mkdtemp(); os.chdir(root)
os.makedirs("img/seq"); os.makedirs("bb/seq")
rng = np.random.default_rng(0)
for i in range(T):
    pair = O.synth_frame_pair(480, 856, seed=5, shift=(1.5 * i, -0.5 * i))
    Image.fromarray((pair[0, 3:, :, :854].permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()).save(f"img/seq/{i:05d}.jpg", quality=95)
    wh = rng.uniform(40, 400, (20, 2)); xy = rng.uniform(0, 1, (20, 2)) * (np.array([854, 480]) - np.minimum(wh, [854, 480]))
    json.dump([{"bbox": [round(float(xy[k, 0]), 1), round(float(xy[k, 1]), 1), round(float(min(wh[k, 0], 854)), 1), round(float(min(wh[k, 1], 480)), 1)],
                "score": 0.9} for k in range(20)], open(f"bb/seq/{i:05d}.json", "w"))
w = RO.synth_weights(0)
eng = rd.RefinementEngine(rd.RefinementNet(w, 16))
rd.forward_directory(eng, "img/", "bb/", "out0/")          # warm: plans built
pr = cProfile.Profile(); pr.enable()
rd.forward_directory(eng, "img/", "bb/", "out1/")
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
