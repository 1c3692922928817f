import os, sys, subprocess, tempfile, pathlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir("/root/repo")
import importlib
tp = importlib.import_module("test_gpu_plumbing")
root = pathlib.Path(tempfile.mkdtemp()) / "g"
root.mkdir()
tp._make_tree(root, t=7)
env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
env.update({"PREMVOS_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONPATH": "/root/repo"})
r = subprocess.run([sys.executable, "-m", "premvos_amd.stream", "--root", str(root), "--gpus", "2", "--flow_weights",
                    "weights/pwc.pth.tar", "--general_weights", "weights/proposal_general_weights", "--specific_weights",
                    "weights/specific.pt", "--refinement_weights", "weights/refinement_specific_weights", "--batch", "2", "--gather"],
                   capture_output=True, text=True, env=env, timeout=1500, cwd="/root/repo")
print(r.returncode); print(r.stdout[-2000:]); print(r.stderr[-6000:])
