#!/usr/bin/env python3
"""Per-kernel register / scratch report of the gfx950 code objects: compiles each csrc/*.hip to device assembly
(hipcc -S, no GPU needed) and prints every kernel whose metadata shows spilled VGPRs / SGPRs or scratch.

    python tools/spill_report.py [file.hip ...] [--all]      (--all: also the kernels without spills)
    python tools/spill_report.py --built [--all]             the same numbers from the SHIPPED objects (vtp_amd/lib/*.o: device code
                                                              object out of .hip_fatbin, AMDGPU metadata note) -- no compile, < 1 s
tests/test_host_logic.py::test_shipped_kernels_spill_ratchet reads the built objects and pins the GEMM family's spill counts."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vtp_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.strip().split("\n")
    except (OSError, subprocess.CalledProcessError):
        return names


def report(path):
    """[(kernel, vgpr_spill, sgpr_spill, scratch_bytes, vgprs, agprs)] of one .hip file"""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        extra = os.environ.get("VTP_HIPCC_EXTRA", "").split()
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
               "--cuda-device-only", "-S", "-o", out, path] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        s = open(out).read()
    rows = []
    for e in s.split("  - .agpr_count:")[1:]:
        g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", e).group(1))
        rows.append([re.search(r"\.name:\s+(\S+)", e).group(1), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                     g("vgpr_count"), int(re.match(r"\s*(\d+)", e).group(1))])
    for row, nm in zip(rows, demangle([r[0] for r in rows])):
        row[0] = nm
    return [tuple(r) for r in rows]


LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _rows_from_metadata(text):
    rows = []
    for e in re.split(r"\n\s*- \.agpr_count:", text)[1:]:
        g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", e).group(1))
        rows.append([re.search(r"\.name:\s+(\S+)", e).group(1), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                     g("vgpr_count"), int(re.match(r"\s*(\d+)", e).group(1))])
    for row, nm in zip(rows, demangle([r[0] for r in rows])):
        row[0] = nm
    return [tuple(r) for r in rows]


def report_built(obj):
    """the same tuples from a built host object (vtp_amd/lib/<file>.o): its .hip_fatbin bundle -> gfx950 code object -> metadata note"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "copy.o")],
                           capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return []  # host-only translation unit (no device code)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    return _rows_from_metadata(notes)


def built_objects():
    lib = os.path.join(ROOT, "vtp_amd", "lib")
    return sorted(os.path.join(lib, f[:-4] + ".o") for f in os.listdir(CSRC) if f.endswith(".hip") and os.path.exists(os.path.join(lib, f[:-4] + ".o")))


def main():
    if "--built" in sys.argv:
        for o in built_objects():
            for name, vs, ss, scr, vg, ag in report_built(o):
                if "--all" in sys.argv or vs or ss or scr:
                    print(f"{os.path.basename(o):24s} vgpr_spill={vs:4d} sgpr_spill={ss:3d} scratch={scr:5d} B  vgpr={vg:3d} agpr={ag:3d}  {name[:140]}")
        return
    show_all = "--all" in sys.argv
    files = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    for f in files:
        for name, vs, ss, scr, vg, ag in report(f):
            if show_all or vs or ss or scr:
                print(f"{os.path.basename(f):24s} vgpr_spill={vs:4d} sgpr_spill={ss:3d} scratch={scr:5d} B  vgpr={vg:3d} agpr={ag:3d}  {name[:140]}")


if __name__ == "__main__":
    main()
