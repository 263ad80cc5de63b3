"""tests/cabi_driver.c — a compiled C caller that replays the ExecFn call sequences of go/arrowhip/register.go against
libarrowhip.so, every step on a pthread of its own (a goroutine hops OS threads between cgo calls, compute/exec.go:165), two
executors at once on two contexts — and this wrapper, which compares every byte it produced with the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi_driver.c")


def build(tmp_path):
    exe = str(tmp_path / "cabi_driver")
    libdir = os.path.join(ROOT, "arrow_go_amd")
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"), SRC, "-L", libdir, "-larrowhip",
           f"-Wl,-rpath,{libdir}", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_cabi_driver_compiles_as_c11_against_the_header(tmp_path):
    """no GPU needed: the driver is plain C11 + pthreads and links against every entry point it names"""
    build(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 70001, 1 << 20])
def test_cabi_driver_matches_oracle(tmp_path, n):
    from tests.backends import OracleBackend
    orc = OracleBackend()
    exe = build(tmp_path)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([exe, str(out), str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi_driver ok" in r.stdout
    rd = lambda name, dt: np.fromfile(out / name, dtype=dt)
    a, b, keys, idx = rd("a.bin", np.int64), rd("b.bin", np.int64), rd("keys.bin", np.int64), rd("idx.bin", np.int32)
    vvalid, fdata, fvalid, ivalid, v0 = (rd(f, np.uint8) for f in ("vvalid.bin", "fdata.bin", "fvalid.bin", "ivalid.bin", "v0.bin"))
    voff, foff = 3, 13
    pad = lambda x: np.concatenate([x, np.zeros(16, np.uint8)])
    # add (scalar kernel, MemPrealloc)
    assert rd("add.bin", np.int64).tobytes() == orc.arithmetic(0, 0, a, b).tobytes() == rd("par_add.bin", np.int64).tobytes()
    # greater at out.Offset = 5: the bytes around the range survive
    before = rd("gt_before.bin", np.uint8)
    got = rd("gt.bin", np.uint8)
    bits = lambda x: np.unpackbits(x, bitorder="little")
    gb, bb = bits(got), bits(before)
    assert (gb[:5] == bb[:5]).all() and (gb[5 + n:] == bb[5 + n:]).all()
    assert (gb[5:5 + n] == (a > 12345)).all()
    pg = bits(rd("par_gt.bin", np.uint8))
    assert (pg[5:5 + n] == (a > 0)).all() and not pg[:5].any()
    # array_filter: Drop / Emit, values with nulls at bit offset 3, filter with nulls at bit offset 13
    for null_sel in (0, 1):
        e = orc.filter(a, pad(vvalid), voff, pad(fdata), pad(fvalid), foff, n, null_sel, True)
        meta = rd(f"filter{null_sel}_meta.bin", np.int64)
        assert meta[0] == len(e[0]) and meta[1] == e[2]
        assert rd(f"filter{null_sel}_vals.bin", np.int64).tobytes() == e[0].tobytes()
        assert rd(f"filter{null_sel}_valid.bin", np.uint8).tobytes() == e[1].tobytes()
    # array_take
    e = orc.take(a, pad(v0), 0, idx, pad(ivalid), 0, True, True)
    assert rd("take_vals.bin", np.int64).tobytes() == e[1].tobytes() and rd("take_valid.bin", np.uint8).tobytes() == e[2].tobytes()
    assert rd("take_meta.bin", np.int64)[0] == e[3]
    rc, msg, bad = open(out / "take_error.txt").read().split("|")
    assert int(rc) == 2 and msg.endswith(f"{bad} out of bounds"), (rc, msg)       # AH_EINDEX, "<v> out of bounds" (helpers.go:950)
    # dictionary_encode (nulls masked)
    e = orc.hash_encode(keys, pad(v0), 0, False)
    assert rd("enc_ids.bin", np.int32).tobytes() == e[0].tobytes() and rd("enc_dict.bin", np.uint64).tobytes() == e[2].tobytes()
    assert (bits(rd("enc_ids_valid.bin", np.uint8))[:n] == bits(e[1])[:n]).all()

