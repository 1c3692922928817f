"""Dev tool: where a short stage-driver run spends its time (model set-up vs frames), via cProfile of the second (warm) call."""
import cProfile, pstats, os, sys, tempfile, json, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from PIL import Image
from oracle import proposal_oracle as PO, pwc_oracle as O
T = 32
root = tempfile.mkdtemp(); os.chdir(root)
sd = "data/DAVIS/JPEGImages/480p/seq"; os.makedirs(sd)
for i in range(T):
    pair = O.synth_frame_pair(480, 856, seed=5, shift=(1.5 * i, -0.5 * i))
    Image.fromarray((pair[0, 3:, :, :854].permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()).save(f"{sd}/{i:05d}.jpg", quality=95)
open("seq_to_run.txt", "w").write("data/DAVIS/JPEGImages/480p/seq/\n")
os.makedirs("weights")
torch.save({"state_dict": O.synth_state_dict(0)}, "weights/pwc.pth.tar")
torch.save(PO.synth_weights(1), "weights/specific.pt")
from premvos_amd.flow import driver as fd
from premvos_amd.proposal import driver as pd
for name, fn in (("flow", lambda: fd.main(["seq_to_run.txt", "weights/pwc.pth.tar", "output/flow"])),
                 ("proposals", lambda: pd.main(["--forward", "output/specific", "--agnostic", "--second_head", "--forward_dataset", "DAVIS",
                                                "--load", "weights/specific.pt", "--davis_name", os.path.join(os.getcwd(), "seq_to_run.txt")]))):
    fn()                                   # cold
    pr = cProfile.Profile(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(18)
    print("=====", name); print("\n".join(l[:150] for l in st.getvalue().splitlines()[:40]))
