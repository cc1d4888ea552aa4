#!/usr/bin/env python3
"""Register / scratch report of every gfx950 kernel in the built objects (csrc/*.o): carves the device ELF out of each object's
.hip_fatbin offload bundle and reads the AMDGPU metadata note (llvm-readelf --notes).
usage: kernel_regs.py [--spills] [substring ...]   (--spills: only kernels with spilled VGPRs or scratch)"""
import glob
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"


def device_elfs(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = data.find(magic)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        p = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                yield data[pos + off:pos + off + size]
        pos = data.find(magic, pos + 1)


def kernels(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]  # noqa: E731
        yield {"name": g("name"), "vgpr": g("vgpr_count"), "agpr": blk.split()[0], "sgpr": g("sgpr_count"), "vgpr_spill": g("vgpr_spill_count"),
               "sgpr_spill": g("sgpr_spill_count"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_spills = "--spills" in sys.argv
    rows = []
    for obj in sorted(glob.glob(os.path.join(ROOT, "three-mlagents_amd", "csrc", "*.o"))):
        for elf in device_elfs(obj):
            for k in kernels(elf):
                k["obj"] = os.path.basename(obj)
                rows.append(k)
    names = subprocess.run([CXXFILT], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        short = re.sub(r"\(.*", "", n).replace("void ", "").replace("tma::", "")
        if args and not any(a in short for a in args):
            continue
        if only_spills and r["vgpr_spill"] in ("0", "?") and r["scratch"] in ("0", "?"):
            continue
        print(f"{r['obj']:16s} vgpr {r['vgpr']:>3s} agpr {r['agpr']:>3s} spill v{r['vgpr_spill']:>3s} s{r['sgpr_spill']:>3s} scratch {r['scratch']:>5s} B  {short}")
