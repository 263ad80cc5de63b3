#!/usr/bin/env python3
"""Per-kernel resources (scratch bytes, VGPRs, SGPRs, LDS) of the gfx950 code objects embedded in libarrowhip.so.
    python scripts/kernel_resources.py [lib] > profiles/rNN_kernel_resources.csv
A kernel with scratch > 0 spills registers or indexes a private array dynamically — neither belongs in a streaming kernel."""
import os, re, struct, subprocess, sys, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "arrow_go_amd", "libarrowhip.so")
data = open(lib, "rb").read()
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
rows = []
pos = 0
while True:
    pos = data.find(b"\x7fELF\x02\x01\x01", pos)
    if pos < 0:
        break
    hdr = data[pos:pos + 64]
    e_machine = struct.unpack_from("<H", hdr, 18)[0]
    if e_machine == 224:  # EM_AMDGPU
        e_shoff, = struct.unpack_from("<Q", hdr, 40)
        e_shentsize, e_shnum = struct.unpack_from("<HH", hdr, 58)
        size = e_shoff + e_shentsize * e_shnum
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(data[pos:pos + size])
        out = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for blk in re.split(r"\n\s+- \.agpr_count:", out)[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, ""])[1]
            rows.append((g("name"), g("private_segment_fixed_size"), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size")))
        pos += max(size, 1)
    else:
        pos += 4
print("kernel,scratch_bytes,vgpr,sgpr,lds_bytes")
for r in sorted(set(rows), key=lambda r: (-int(r[1] or 0), r[0])):
    print(",".join(r))
