"""The nontemporal hints of the streaming kernels are checked in the ISA (no GPU needed: hipcc cross-compiles gfx950).

They have been lost silently twice: `__builtin_nontemporal_load` through a `__builtin_bit_cast` to the carrier struct compiled to a plain
load, and a hinted and a plain access of one address and type behind an inlined switch were merged into the plain one — the clustered
Take ran 12 % slower each time and every parity test stayed green.  This test disassembles the kernels and looks for `nt` on their 16-byte
accesses — and, while the disassembly is there, checks the register / scratch budgets whose violation costs a kernel
its second workgroup per CU."""
import os
import re
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "arrow_go_amd", "csrc")

# source → [(mangled-name regex, {instruction-with-hint: minimum count})]
EXPECT = {
    "ah_take.hip": [
        (r"take_vec_kernelILi8EiLb0ELi7E", {"global_load_dwordx4 nt": 4, "global_store_dwordx4 nt": 4, "global_load_dwordx2 nt": 4}),
        (r"take_vec_kernelILi8EiLb1ELi7E", {"global_load_dwordx4 nt": 4, "global_store_dwordx4 nt": 4}),
        (r"take_kernelILi8EiLb0ELb1E", {"global_load_dword nt": 1, "global_store_dwordx2 nt": 1}),
    ],
    "ah_filter.hip": [
        (r"compact_kernelILi8ELb1ELb0ELb1E", {"global_load_dwordx4 nt": 4, "global_store_dwordx4 nt": 1}),
        (r"compact_kernelILi4ELb1ELb0ELb1E", {"global_load_dwordx4 nt": 4, "global_store_dwordx4 nt": 1}),
    ],
    "ah_ctx.hip": [(r"copy16_kernel", {"global_load_dwordx4 nt": 4, "global_store_dwordx4 nt": 4})],   # the measured ceiling; ah_copy_async
    "ah_sum.hip": [
        (r"sum_partials_kernelIdN\w*AccDDELb1E", {"global_load_dwordx4 nt": 4}),
        (r"sum_partials_kernelImN\w*AccU64ELb1E", {"global_load_dwordx4 nt": 4}),
    ],
    "ah_compare.hip": [(r"compare_kernelIlLi2ELi1ELb1ELb1E", {"global_load_dwordx4 nt": 4})],          # greater(Int64 array, scalar)
    "ah_fused.hip": [(r"fused_kernelIlLi2ELb0ELb1E", {"global_load_dwordx4 nt": 4})],                   # C4: Compare(>) → Filter → Sum
    "ah_arith.hip": [
        (r"binary_kernelImLi0ELi0ELb1ELb1E", {"global_load_dwordx4 nt": 8, "global_store_dwordx4 nt": 4}),   # aligned Int64 Add: the headline kernel
        (r"binary_kernelImLi0ELi0ELb0ELb1E", {"global_load_dwordx4 nt": 8, "global_store_dwordx4 nt": 4}),   # … over an element-aligned slice
    ],
}


# source → [(mangled-name regex, max VGPRs or None, max scratch bytes)]: the occupancy cliffs of DESIGN.md §8 — a 1024-thread kernel that runs two
# workgroups per CU has 64 registers (the group-by scatter at 66: 0.44 → 0.64 ms), 256-thread kernels with 8 waves per SIMD the same
RESOURCES = {
    "ah_groupby.hip": [
        (r"gb_scatter_kernelILb1E", 64, 0),
        (r"gb_aggregate_kernelILb1ELb0E", 128, 0),
        # the two-level cut's scatter (round 6): at 68–71 registers one workgroup of 1024 per CU — level 2 ran at 3.5 TB/s, 772 µs; at 64: 660
        (r"gs_scatter_kernelINS_9GsRecordsE", 64, 16),
        (r"gbr_emit_kernel", 64, 0),
    ],
    "ah_hash_part.hip": [
        (r"gb_scatter_kernelILb0E", 64, 0),
        (r"enc_unpermute_group_kernelILi4E", 64, 0),
        (r"e2_unpermute_group_kernelILi4E", 64, 0),
        (r"enc_table_kernelILi8192ELb1E", 128, 0),
    ],
    "ah_take.hip": [(r"take_vec_kernelILi8EiLb[01]ELi7E", 64, 0)],
    # the one-pass cumulative_sum: a workgroup of 1024 lanes per CU = 128 registers, and nothing of the tile in scratch (a CSE of the
    # sixteen null masks across the look-back once spilled 21 … 49 registers and every parity test stayed green: DESIGN.md §3.4)
    "ah_scan.hip": [(r"scan_onepass_kernelI[jyt]Lb", 128, 16), (r"scan_onepass_f64_kernel", 128, 0)],
}


def _disassemble(src, outdir):
    out = os.path.join(outdir, src + ".s")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-S",
                        "--cuda-device-only", "-o", out, os.path.join(CSRC, src)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    mix, cur = {}, None
    res = {}
    text = open(out).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        p = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        res[m.group(1)] = (int(v.group(1)) if v else 0, int(p.group(1)) if p else 0)
    for line in text.splitlines(keepends=True):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            mix[cur] = {}
            continue
        if cur and line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur:
            m = re.search(r"\b(global_(?:load|store)_\w+)\b", line)
            if m:
                k = m.group(1) + (" nt" if re.search(r"\bnt\b", line) else "")
                mix[cur][k] = mix[cur].get(k, 0) + 1
    return mix, res


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_streaming_kernels_carry_their_nontemporal_hints_and_stay_inside_their_register_budgets():
    sources = sorted(set(EXPECT) | set(RESOURCES))
    with tempfile.TemporaryDirectory() as d:
        with ThreadPoolExecutor(len(sources)) as ex:
            parsed = dict(zip(sources, ex.map(lambda s: _disassemble(s, d), sources)))
    for src, checks in EXPECT.items():
        mixes = parsed[src][0]
        for pat, want in checks:
            hits = [k for k in mixes if re.search(pat, k)]
            assert hits, f"{src}: no kernel matches {pat}"
            for k in hits:
                for ins, n in want.items():
                    assert mixes[k].get(ins, 0) >= n, f"{src}: {k}: expected ≥ {n} × '{ins}', ISA has {mixes[k]}"
    for src, checks in RESOURCES.items():
        res = parsed[src][1]
        for pat, max_vgpr, max_scratch in checks:
            hits = [k for k in res if re.search(pat, k)]
            assert hits, f"{src}: no kernel matches {pat}"
            for k in hits:
                vgpr, scratch = res[k]
                assert vgpr <= max_vgpr, f"{src}: {k}: {vgpr} VGPRs, budget {max_vgpr}"
                assert scratch <= max_scratch, f"{src}: {k}: {scratch} bytes of scratch, budget {max_scratch}"
