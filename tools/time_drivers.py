"""Dev tool: file-to-file throughput of the stage drivers (JPEG decode, .flo / JSON / RLE included) on a synthetic 480p sequence
with full-depth nets:  python tools/time_drivers.py [frames=24]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from oracle import proposal_oracle as PO, pwc_oracle as O, refinement_oracle as RO, reid_oracle as QO
T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
REPO = os.getcwd()
root = tempfile.mkdtemp()
os.chdir(root)
sd = "data/DAVIS/JPEGImages/480p/seq"
os.makedirs(sd)
for i in range(T):
    pair = O.synth_frame_pair(480, 856, seed=5, shift=(1.5 * i, -0.5 * i))
    Image.fromarray((pair[0, 3:, :, :854].permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()).save(f"{sd}/{i:05d}.jpg", quality=95)
open("seq_to_run.txt", "w").write("data/DAVIS/JPEGImages/480p/seq/\n")
os.makedirs("weights")
torch.save({"state_dict": O.synth_state_dict(0)}, "weights/pwc.pth.tar")
torch.save(PO.synth_weights(0), "weights/general.pt"); torch.save(PO.synth_weights(1), "weights/specific.pt")
torch.save(RO.synth_weights(0), "weights/refine.pt"); torch.save(QO.synth_weights(0), "weights/reid.pt")
os.makedirs("code/ReID_net/configs")
json.dump({"model": "Re-ID", "load": os.path.abspath("weights/reid.pt")}, open("code/ReID_net/configs/run", "w"))
# time each stage by running the pipeline and watching directory creation order through a patched print
t0 = time.time()
stamps = {}
orig = os.path.isdir
def stage_time(name, fn):
    torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); stamps[name] = time.time() - t
from premvos_amd.flow import driver as fd
from premvos_amd.proposal import driver as pd
from premvos_amd.proposal.combine import combine
from premvos_amd.refinement import driver as rd
from premvos_amd.reid import driver as qd
inter = "output/intermediate"
ONLY_STREAM = os.environ.get("TIME_DRIVERS_ONLY_STREAM") == "1"     # dev: skip the stage drivers
for rep in (() if ONLY_STREAM else ("cold", "warm")):            # cold = plan building + autotuning included; warm = second sequence-equivalent
    os.system(f"rm -rf {root}/output")
    stage_time(f"flow/{rep}", lambda: fd.main(["seq_to_run.txt", "weights/pwc.pth.tar", f"{inter}/flow"]))
    for name, wf in (("general_proposals", "weights/general.pt"), ("specific_proposals", "weights/specific.pt")):
        stage_time(f"{name}/{rep}", lambda: pd.main(["--forward", f"{inter}/{name}", "--agnostic", "--second_head", "--forward_dataset", "DAVIS",
                                                     "--load", wf, "--davis_name", os.path.join(os.getcwd(), "seq_to_run.txt")]))
    combine(inter + "/")
    if rep == "cold":
        w = rd.load_weights("weights/refine.pt"); r_eng = rd.RefinementEngine(rd.RefinementNet(w, rd.infer_num_middle(w)))
        q_eng = qd.engine_from_config(qd.Config("code/ReID_net/configs/run"))
    stage_time(f"refinement/{rep}", lambda: rd.forward_directory(r_eng, "data/DAVIS/JPEGImages/480p/", f"{inter}/combined_proposals/", f"{inter}/refined_proposals/"))
    stage_time(f"reid/{rep}", lambda: qd.forward_directory(q_eng, "data/DAVIS/JPEGImages/480p/", f"{inter}/refined_proposals/", f"{inter}/ReID_proposals/"))
if ONLY_STREAM:
    r_eng = q_eng = None
    stamps.update({f"{k}/warm": float("nan") for k in ("flow", "general_proposals", "specific_proposals", "refinement")})
# the optional binary side-car (PREMVOS_SIDECAR=1): refinement writes <frame>.pmv (bit-packed masks), ReID reads it
if not ONLY_STREAM: stage_time("refinement(side-car)/warm", lambda: rd.forward_directory(r_eng, "data/DAVIS/JPEGImages/480p/", f"{inter}/combined_proposals/", f"{inter}/refined_sidecar/", sidecar=True))
if not ONLY_STREAM: stage_time("reid(side-car)/warm", lambda: qd.forward_directory(q_eng, "data/DAVIS/JPEGImages/480p/", f"{inter}/refined_sidecar/", f"{inter}/ReID_sidecar/"))
# the optional GPU JPEG decode (PREMVOS_GPU_JPEG=1): the pool only Huffman-decodes, inverse DCT / colour conversion on the GPU
os.environ["PREMVOS_GPU_JPEG"] = "1"
os.system(f"rm -rf {root}/{inter}/flow")
if not ONLY_STREAM: stage_time("flow(gpu-jpeg)/warm", lambda: fd.main(["seq_to_run.txt", "weights/pwc.pth.tar", f"{inter}/flow"]))
os.environ["PREMVOS_GPU_JPEG"] = "0"
del r_eng, q_eng                          # (their plans hold tens of GB of activations; the streaming pipeline builds its own)
import gc; gc.collect(); torch.cuda.empty_cache()
# the four hot-path stages as ONE streaming process (one decode per frame, stages overlapped on three host threads)
from premvos_amd import stream
sp = stream.StreamPipeline("weights/pwc.pth.tar", "weights/general.pt", "weights/specific.pt", "weights/refine.pt",
                           batch=int(os.environ.get("PREMVOS_STREAM_BATCH", "8")))
for rep in ("cold", "warm", "warm2"):    # cold = plans built on the fly
    os.system(f"rm -rf {root}/output")
    stage_time(f"stream A+B+B+C+D/{rep}", lambda: sp.run_sequences(["data/DAVIS/JPEGImages/480p/seq/"]))
os.environ["PREMVOS_GPU_JPEG"] = "1"
for rep in ("warm", "warm2"):
    os.system(f"rm -rf {root}/output")
    stage_time(f"stream(gpu-jpeg)/{rep}", lambda: sp.run_sequences(["data/DAVIS/JPEGImages/480p/seq/"]))
os.environ["PREMVOS_GPU_JPEG"] = "0"
nprops = sum(len(json.load(open(f"{root}/{inter}/combined_proposals/seq/{i:05d}.json"))) for i in range(T)) / T
print(f"{T} frames 480x854, {nprops:.1f} combined proposals per frame; DRIVER_BATCH={os.environ.get('PREMVOS_DRIVER_BATCH', 'default')}")
for k, v in stamps.items():
    print(f"  {k:28s} {v:7.2f} s  = {T / v:6.1f} frames/s")
serial = sum(stamps[f"{k}/warm"] for k in ("flow", "general_proposals", "specific_proposals", "refinement"))
summary = {"frames": T, "proposals_per_frame": round(nprops, 1),
           "stage_drivers_one_after_the_other_fps": round(T / serial, 2), "streaming_driver_fps": round(T / min(stamps["stream A+B+B+C+D/warm"], stamps["stream A+B+B+C+D/warm2"]), 2),
           "streaming_driver_gpu_jpeg_fps": round(T / min(stamps["stream(gpu-jpeg)/warm"], stamps["stream(gpu-jpeg)/warm2"]), 2),
           "per_stage_fps": {k.split("/")[0]: round(T / v, 1) for k, v in stamps.items() if k.endswith("/warm")}}
print(json.dumps(summary))
if len(sys.argv) > 2:
    json.dump(summary, open(os.path.join(REPO, sys.argv[2]), "w"), indent=1)
