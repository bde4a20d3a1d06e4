#!/usr/bin/env python3
"""Per-kernel resources of the built library (registers, spills, scratch, static LDS) from the gfx950 code object's metadata notes.

    python tools/kernel_resources.py [pattern ...]        # table of the kernels whose name contains one of the patterns (default: all with scratch)

Used by tests/test_host_logic.py::test_hot_kernels_do_not_spill (a compiler update that makes project_x3_stream_kernel spill must
fail the CPU suite, not show up as a slower bench).  Reads tgcn_amd/lib/libtgcn_hip.so: .hip_fatbin -> clang offload bundle -> the
hipv4-amdgcn-amd-amdhsa--gfx950 ELF -> `llvm-readelf --notes` (AMDGPU metadata, YAML-like)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
        "max_flat_workgroup_size", "uses_dynamic_stack")


def code_object(so_path, arch="gfx950"):
    """bytes of the device ELF for `arch` inside the shared library"""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        # (an explicit output file: objcopy with ONE file argument rewrites its input in place -- round 6 found the product library 600 KB larger and
        #  re-stamped after every CPU test run)
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path, os.path.join(d, "copy.so")], check=True)
        b = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    assert b.startswith(magic), "not a clang offload bundle"
    n = struct.unpack_from("<Q", b, len(magic))[0]
    off = len(magic) + 8
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", b, off)
        off += 24
        triple = b[off: off + tl].decode()
        off += tl
        if arch in triple:
            return b[o: o + sz]
    raise RuntimeError("no %s code object in %s" % (arch, so_path))


def kernel_resources(so_path=None):
    """{demangled-ish kernel name: {key: int}} for every kernel of the library"""
    so_path = so_path or os.path.join(ROOT, "tgcn_amd", "lib", "libtgcn_hip.so")
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "dev.co")
        open(co, "wb").write(code_object(so_path))
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    res, cur = {}, None
    for ln in notes.splitlines():
        if re.match(r"\s+- \.(agpr_count|args):", ln):          # first key of a kernel's map (keys are sorted)
            if cur and "name" in cur:
                res[cur["name"]] = cur
            cur = {}
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", ln)
        if cur is not None and m:
            k, v = m.group(1), m.group(2)
            if k == "name" and v.startswith("_Z"):
                cur["name"] = v
            elif k in KEYS:
                cur[k] = int(v) if v.lstrip("-").isdigit() else v
    if cur and "name" in cur:
        res[cur["name"]] = cur
    return res


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            out = subprocess.run([tool] + list(names), capture_output=True, text=True)
        except FileNotFoundError:
            continue
        if out.returncode == 0:
            return out.stdout.splitlines()
    return list(names)


if __name__ == "__main__":
    pats = sys.argv[1:]
    res = kernel_resources()
    names = sorted(res)
    pretty = dict(zip(names, demangle(names)))
    print("%d kernels" % len(res))
    for nm in names:
        r = res[nm]
        scratch = r.get("private_segment_fixed_size", 0) or r.get("vgpr_spill_count", 0) or r.get("sgpr_spill_count", 0)
        if (pats and any(p in pretty[nm] for p in pats)) or (not pats and scratch):
            print("%-110s vgpr %3s agpr %3s sgpr %3s  spill v %3s s %3s  scratch %4s B  lds %6s B" % (
                pretty[nm].replace("(anonymous namespace)::", "")[:110], r.get("vgpr_count"), r.get("agpr_count"), r.get("sgpr_count"),
                r.get("vgpr_spill_count"), r.get("sgpr_spill_count"), r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
