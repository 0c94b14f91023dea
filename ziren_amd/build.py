"""Build libzkm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "zkm_hip.hip")
OUT = os.path.join(HERE, "libzkm_hip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))] + [
    os.path.join(os.path.dirname(HERE), "include", "zkm_hip.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


LAST = {"compiled": None}      # what the last build() call did: True = hipcc ran, False = the in-tree .so was newer than every source


def build(force=False, verbose=False):
    if not force and not needs_build():
        LAST["compiled"] = False
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wl,-rpath,/opt/rocm/lib",
           SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    LAST["compiled"] = True
    return OUT


def hipcc_version():
    try:
        out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True, timeout=60).stdout
        return next((l.strip() for l in out.splitlines() if "HIP version" in l), out.splitlines()[0].strip() if out else None)
    except (OSError, subprocess.SubprocessError):
        return None


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
