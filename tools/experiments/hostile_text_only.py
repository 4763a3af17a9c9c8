"""Only the hostile-text family of tools/fuzz_adapters.py (device MOTION parser on truncated / flipped / random bytes), printing every block the host
parser rejects and the device neither flags nor hands over.  Found (round 3): extra tokens on the LAST row asked for went unnoticed."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import _native
import tests.test_gpu_adapters as T
lib = _native.load(); dev = torch.device("cuda", 0); vp = T.vp
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
found = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20000):
    n_cols = int(rng.integers(1, 12)); n_lines = int(rng.integers(1, 40))
    base = "\n".join(" ".join("%.6f" % v for v in rng.normal(0, 30, n_cols)) for _ in range(n_lines)).encode() + b"\n"
    b2 = bytearray(base)
    kind = int(rng.integers(0, 5))
    if kind == 0:
        b2 = b2[: int(rng.integers(0, len(b2) + 1))]
    elif kind == 1:
        for _k in range(int(rng.integers(1, 6))):
            b2[int(rng.integers(0, len(b2)))] = int(rng.integers(0, 256))
    elif kind == 2:
        alphabet = b"0123456789.eE+- \t\r\n" + bytes([0, 255, ord("x"), ord("n"), ord("a")])
        b2 = bytearray(alphabet[int(i)] for i in rng.integers(0, len(alphabet), int(rng.integers(1, 9000))))
    elif kind == 3:
        toks = bytes(b2).split()
        toks[int(rng.integers(0, len(toks)))] = [b"1e999", b"9" * 40, b"1e", b"-", b".", b"nan", b"inf", b"0x10", b"1e-400", b"0." + b"0" * 60 + b"7", b"+.5e+2"][int(rng.integers(0, 11))]
        b2 = bytearray(b" ".join(toks))
    text = bytes(b2)
    if not text: continue
    junk = b"#" * int(rng.integers(0, 70))
    blob = junk + text
    rc, rows_d, status, ntok, slow, ns = T._device_parse(lib, dev, blob, [(len(junk), len(junk) + len(text))], [n_lines], n_cols, max_slow=1 << 14)
    hout = np.full(n_lines * n_cols + 8, np.nan); nl_h, nc_h = C.c_int64(0), C.c_int64(0)
    got_h = lib.gmr_bvh_parse_motion(text, len(text), n_lines, hout.ctypes.data_as(vp), n_lines * n_cols, C.byref(nl_h), C.byref(nc_h))
    acc = got_h == n_lines * n_cols and nc_h.value == n_cols
    flagged = status[0] != 0 or ns > 0 or int(ntok[0]) < n_lines * n_cols
    if not acc and not flagged:
        found += 1
        print("UNFLAGGED", "kind", kind, "n_cols", n_cols, "n_lines", n_lines, "host", got_h, nl_h.value, nc_h.value, "ntok", int(ntok[0]), "text", text[:300])
        if found > 5: break
print("done: unflagged rejects", found)
