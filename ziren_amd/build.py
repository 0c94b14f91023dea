"""Build libzkm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Whether the in-tree library is current is decided by content, not by time: the sha256 of every file of csrc/ and of include/zkm_hip.h
(`sources_digest`) is compiled into the library (`zkm_build_info()`, a string that is also readable from the file's bytes) and compared
with the tree's. A source edited with an older mtime, or a library built from another checkout, is rebuilt; an unchanged tree proves it
needs no compilation by showing the same digest."""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "zkm_hip.hip")
OUT = os.path.join(HERE, "libzkm_hip.so")
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if not f.startswith(".")) + [
    os.path.join(os.path.dirname(HERE), "include", "zkm_hip.h")]
_MARK = re.compile(rb"ZKM_SOURCES_DIGEST=([0-9a-f]{16});")


def sources_digest():
    """sha256 (16 hex digits) over the library's sources: every file of ziren_amd/csrc in name order, then include/zkm_hip.h."""
    h = hashlib.sha256()
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def recorded_digest(path=None):
    """The digest a built library carries (None: no library, or one built before digests were recorded). Read from the file's bytes, so
    checking it does not load the library."""
    path = path or OUT
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = _MARK.search(f.read())
    return m.group(1).decode() if m else None


def needs_build():
    return recorded_digest() != sources_digest()


LAST = {"compiled": None, "why": None}      # what the last build() call did: True = hipcc ran; False = the library's digest is the tree's


def build(force=False, verbose=False):
    want = sources_digest()
    have = recorded_digest()
    if not force and have == want:
        LAST.update(compiled=False, why=f"libzkm_hip.so carries the tree's sources digest {want}")
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    version = (hipcc_version() or "unknown").replace('"', "'")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wl,-rpath,/opt/rocm/lib",
           f'-DZKM_SOURCES_DIGEST="{want}"', f'-DZKM_HIPCC_VERSION="{version}"', SRC, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    if recorded_digest() != want:
        raise RuntimeError("the library just built does not carry the sources digest it was given")
    LAST.update(compiled=True, why="forced" if force and have == want else f"library digest {have} != tree digest {want}")
    return OUT


def hipcc_version():
    try:
        out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True, timeout=60).stdout
        return next((l.strip() for l in out.splitlines() if "HIP version" in l), out.splitlines()[0].strip() if out else None)
    except (OSError, subprocess.SubprocessError):
        return None


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LAST, file=sys.stderr)
