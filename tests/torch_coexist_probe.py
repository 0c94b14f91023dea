"""GPU-box probe: does libzkm_hip.so work in a process that also has torch (bundled HIP runtime) loaded?"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
order = sys.argv[1] if len(sys.argv) > 1 else "both"
if order == "both":
    for o in ("torch_first", "lib_first"):
        r = subprocess.run([sys.executable, __file__, o], capture_output=True, text=True)
        print(o, "rc", r.returncode, (r.stdout + r.stderr)[-600:].replace("\n", " | "))
    sys.exit(0)
if order == "torch_first":
    import torch
    x = torch.ones(1024, device="cuda") * 2
    torch.cuda.synchronize()
import __graft_entry__ as g
g.smoke()
if order == "lib_first":
    import torch
    x = torch.ones(1024, device="cuda") * 2
    torch.cuda.synchronize()
    print("torch after lib ok", float(x.sum()))
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l})
print("hip runtimes mapped:", libs)
