#!/usr/bin/env python3
"""Which kernels of libtgsr_hip.so contain packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)?

profiles/HISTORY.md 3.13: such an instruction can read registers a following load has already overwritten while other waves' MFMAs keep
the matrix pipe busy, so the inference path is built without them (csrc/Makefile).  This tool pulls the gfx950 code objects out of
the library's clang offload bundles, disassembles them with llvm-objdump and prints kernel -> count.
    python3 tools/scan_packed_fp32.py [--strict] [path/to/lib.so]
The list goes to stdout.  --strict (what `make` runs after linking): exit code 1 if any kernel contains one, or if the scan
did not see device code at all (no code object / no MFMA in the largest one) - a build with such an instruction fails."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")


def code_objects(path):
    """Every device code object (bytes) of the offload bundles embedded in `path`."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return out
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tsz = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tsz].decode()
            p += 24 + tsz
            if "amdgcn" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos += len(MAGIC)


def scan(path):
    """{kernel symbol: number of packed fp32 instructions} for every kernel that has any."""
    counts = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            asm = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
            elif cur and PACKED.search(line):
                counts[cur] = counts.get(cur, 0) + 1
    return counts


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--strict"]
    strict = "--strict" in sys.argv[1:]
    lib = args[0] if args else os.path.join(ROOT, "tgsr_amd", "lib", "libtgsr_hip.so")
    res = scan(lib)
    for k, v in sorted(res.items()):
        print("%6d  %s" % (v, k))
    print("%d kernels with packed fp32 instructions" % len(res))
    if strict:
        cos = code_objects(lib)
        if not cos:
            sys.exit("scan_packed_fp32: no device code object found in %s" % lib)
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(max(cos, key=len))
            f.flush()
            if "v_mfma_f32" not in subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout:
                sys.exit("scan_packed_fp32: the disassembly shows no MFMA - not looking at the kernels")
        if res:
            sys.exit("scan_packed_fp32: packed fp32 instructions in the library (profiles/HISTORY.md 3.13) - build refused")
