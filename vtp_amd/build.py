"""Builds libvtp_hip.so (hand-written gfx950 HIP kernels + the C ABI of include/vtp_hip.h) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the authoring container; the resulting .so travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.environ.get("VTP_BUILD_DIR") or os.path.join(HERE, "lib")  # VTP_BUILD_DIR: a second build (A/B of -D switches, loaded with VTP_HIP_LIB)
LIB = os.path.join(LIBDIR, "libvtp_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-ffp-contract=on", "-I", CSRC, "-I",
         os.path.join(ROOT, "include")] + os.environ.get("VTP_HIPCC_EXTRA", "").split()  # e.g. -DVTP_P8_FOUR_PHASE (A/B builds)


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    for f in _sources() + hdrs + [os.path.join(ROOT, "include", "vtp_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(f for f in FLAGS if not os.path.isabs(f)).encode())  # not the checkout path: the GPU box must not rebuild
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libvtp_hip.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB

    def cc(src):
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, _sources()))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"[vtp_amd.build] built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
