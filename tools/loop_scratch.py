#!/usr/bin/env python3
"""Register spills where they would hurt: scratch instructions inside the innermost MFMA loops (the K loops) of the GEMM and
attention kernels, read from the device code of the BUILT library (ssak_amd/lib/libssak_hip.so: the clang offload bundles of
its .hip_fatbin section -> llvm-objdump).  The persistent GEMMs carry a few spilled registers in their per-tile prologue /
epilogue blocks (hipcc -Rpass-analysis reports them per kernel, tools/kernel_resources.py); those cost nothing measurable.  One
more live register in the K loop and the same report would look no different -- this tool tells the two apart.

    python tools/loop_scratch.py [name filter regex]       prints, per kernel: scratch instructions total / inside K loops
tests/test_abi.py::test_no_scratch_inside_the_matrix_loops asserts on `check()`."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ssak_amd", "lib", "libssak_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
HOT = r"gemm_p4_kernel|gemm_p8_kernel|attn_fwd_kernel|attn_bwd_dq_kernel|attn_bwd_dkv_kernel"


def code_objects(lib=LIB):
    """The gfx950 ELF images of every translation unit in the library's fat binary."""
    data = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    i = 0
    while True:
        i = data.find(magic, i)
        if i < 0:
            return
        (n,) = struct.unpack_from("<Q", data, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + ts].decode()
            p += ts
            if "gfx950" in triple and size:
                yield data[i + off:i + off + size]
        i += len(magic)


def kernels(lib=LIB, pattern=HOT):
    """{mangled kernel name: [(address, text), ...]} of the kernels whose name matches `pattern`."""
    out = {}
    rx = re.compile(pattern)
    for blob in code_objects(lib):
        if not rx.search(blob.decode("latin1")):
            continue
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^([0-9a-f]+) <(\S+)>:$", line)
            if m:
                cur = out.setdefault(m.group(2), []) if rx.search(m.group(2)) else None
                continue
            if cur is None:
                continue
            m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
            if m:
                cur.append((int(m.group(2), 16), m.group(1)))
    return out


def matrix_loops(insts):
    """Innermost loops that contain MFMA instructions: (first, last) instruction indices.  A loop = a backward branch and its target."""
    addr_to_idx = {a: i for i, (a, _) in enumerate(insts)}
    loops = []
    for i, (a, t) in enumerate(insts):
        m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", t)
        if not m:
            continue
        simm = int(m.group(1) or m.group(2))
        if simm >= 0x8000:
            simm -= 0x10000
        tgt = a + 4 + 4 * simm
        if tgt <= a and tgt in addr_to_idx:
            loops.append((addr_to_idx[tgt], i))
    has_mfma = lambda lo, hi: any("v_mfma" in insts[k][1] for k in range(lo, hi + 1))
    mm = [l for l in loops if has_mfma(*l)]
    return [l for l in mm if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in mm)]


def check(lib=LIB, pattern=HOT):
    """[(kernel, scratch instructions in the whole kernel, scratch instructions inside its matrix loops, matrix loops)]"""
    rows = []
    for name, insts in sorted(kernels(lib, pattern).items()):
        loops = matrix_loops(insts)
        total = sum("scratch_" in t for _, t in insts)
        inside = sum("scratch_" in insts[k][1] for lo, hi in loops for k in range(lo, hi + 1))
        rows.append((name, total, inside, len(loops)))
    return rows


if __name__ == "__main__":
    rows = check(pattern=sys.argv[1] if len(sys.argv) > 1 else HOT)
    names = subprocess.run(["/usr/bin/c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':100s} {'scratch':>8s} {'in K loops':>10s} {'K loops':>8s}")
    for (n, total, inside, nl), dn in zip(rows, names):
        dn = re.sub(r"\(anonymous namespace\)::|^void ", "", dn)
        dn = re.sub(r"\(.*$", "", dn)
        print(f"{dn[:100]:100s} {total:8d} {inside:10d} {nl:8d}")
