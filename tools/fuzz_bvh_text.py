#!/usr/bin/env python3
"""Robustness fuzz of the HOST text entry points (no GPU needed): gmr_bvh_parse_header and gmr_bvh_parse_motion on mutated BVH text.

    python tools/fuzz_bvh_text.py [seconds] [seed]

Every call gets its text in a buffer that ENDS at a PROT_NONE guard page (a read one byte past `len` is a segmentation fault, not a
silent success) and output arrays sized exactly as declared with canary words on both sides.  Mutations of the golden files: truncation
at any byte, byte flips, token deletion / duplication / replacement by hostile tokens (huge numbers, empty braces, 300-digit
mantissas, NUL bytes), joint counts around the capacity.  Checked: no crash, canaries intact, return codes in range, reported sizes within
the capacities passed in, and -- for the unmutated files -- the same header the tests pin.
"""
import ctypes as C
import glob
import mmap
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gmr_amd import _native  # noqa: E402

PAGE = mmap.PAGESIZE
libc = C.CDLL(None, use_errno=True)
libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]


class Guarded:
    """`n` bytes that end exactly at an inaccessible page."""

    def __init__(self, cap):
        self.pages = (cap + PAGE - 1) // PAGE + 1
        self.m = mmap.mmap(-1, self.pages * PAGE)
        self.base = C.addressof(C.c_char.from_buffer(self.m))
        if libc.mprotect(self.base + (self.pages - 1) * PAGE, PAGE, 0) != 0:
            raise OSError(C.get_errno(), "mprotect")
        self.cap = (self.pages - 1) * PAGE

    def put(self, data: bytes) -> int:
        assert len(data) <= self.cap
        off = self.cap - len(data)
        self.m[off:self.cap] = data
        return self.base + off


CANARY = 0x5AD0BEEF5AD0BEEF


def canaried(nbytes):
    """(array with 64 canary bytes either side, address of the payload, check function)."""
    pad = 64
    a = np.empty(pad + nbytes + pad, np.uint8)
    a[:pad].view(np.uint64)[:] = CANARY
    a[pad + nbytes:].view(np.uint64)[:] = CANARY if (nbytes % 8 == 0) else CANARY
    tail = a[pad + nbytes:].copy()

    def ok():
        return bool((a[:pad].view(np.uint64) == CANARY).all() and np.array_equal(a[pad + nbytes:], tail))
    return a, a.ctypes.data + pad, ok


HOSTILE = [b"1e999", b"-1e-999", b"9" * 300, b"0." + b"0" * 400 + b"1", b"{", b"}", b"JOINT", b"End", b"Site", b"OFFSET", b"CHANNELS", b"9", b"10", b"-1",
           b"Xrotation", b"Yposition", b"Zscale", b"nan", b"inf", b"0x1p3", b"1e", b"+", b"-", b".", b"\x00", b"\xff\xfe", b"MOTION", b"Frames:", b"Frame", b"Time:",
           b"18446744073709551616", b"999999999999999999", b"a" * 5000]


def mutate(rng, data: bytes) -> bytes:
    k = int(rng.integers(0, 7))
    if k == 0:
        return data[: int(rng.integers(0, len(data) + 1))]
    if k == 1:
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 8))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        return bytes(b)
    toks = data.split()
    if not toks:
        return data
    i = int(rng.integers(0, min(len(toks), 400)))
    if k == 2:
        del toks[i]
    elif k == 3:
        toks.insert(i, toks[i])
    elif k == 4:
        toks[i] = HOSTILE[int(rng.integers(0, len(HOSTILE)))]
    elif k == 5:
        toks.insert(i, HOSTILE[int(rng.integers(0, len(HOSTILE)))])
    else:  # nest joints deeply / widely
        toks[i:i] = [b"JOINT", b"j%d" % int(rng.integers(0, 1000)), b"{", b"OFFSET", b"0", b"0", b"0", b"CHANNELS", b"3", b"Zrotation", b"Yrotation", b"Xrotation"] * int(rng.integers(1, 300))
    sep = [b" ", b"\n", b"\t", b"\r\n"][int(rng.integers(0, 4))]
    return sep.join(toks)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib = _native.load()
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.bvh")))
    seeds = [open(f, "rb").read() for f in files]
    guard = Guarded(4 << 20)
    t0 = time.time()
    runs = bad = parsed = rows_ok = 0
    while time.time() - t0 < seconds:
        src = seeds[int(rng.integers(0, len(seeds)))]
        head_end = src.find(b"Frame Time:")
        head_end = src.find(b"\n", head_end) + 1
        base = src[: head_end + int(rng.integers(0, 4000))]   # the header and a few motion rows
        u = rng.random()
        if u < 0.05:
            data = base
        elif u < 0.5:   # header intact, hostile motion rows: the number parser's turn
            data = base[:head_end] + mutate(rng, base[head_end:] or b"0")
        else:
            data = mutate(rng, base)
        if len(data) > guard.cap:
            data = data[: guard.cap]
        addr = guard.put(data)
        maxj = int(rng.choice([1, 2, 21, 22, 23, 101, 256]))
        ncap = int(rng.choice([1, 16, 64 * maxj]))
        names, p_names, ok_names = canaried(ncap)
        parents, p_par, ok_par = canaried(4 * maxj)
        offsets, p_off, ok_off = canaried(24 * maxj)
        chans, p_ch, ok_ch = canaried(4 * maxj)
        order, p_ord, ok_ord = canaried(12)
        nf, ft, mo = C.c_int64(-7), C.c_double(-7.0), C.c_size_t(0)
        rc = lib.gmr_bvh_parse_header(C.c_void_p(addr), len(data), maxj, C.c_void_p(p_names), ncap, C.c_void_p(p_par), C.c_void_p(p_off), C.c_void_p(p_ch),
                                      C.c_void_p(p_ord), C.byref(nf), C.byref(ft), C.byref(mo))
        runs += 1
        good = all(f() for f in (ok_names, ok_par, ok_off, ok_ch, ok_ord)) and (rc in (-1, -2) or 1 <= rc <= maxj)
        if rc > 0:
            parsed += 1
            good = good and mo.value <= len(data)
            par = parents[64:64 + 4 * rc].view(np.int32)
            good = good and par[0] == -1 and all(0 <= par[i] < i for i in range(1, rc))
            # the motion rows behind it, into a buffer of a random capacity
            max_out = int(rng.choice([0, 1, 7, 500, 5000]))
            out, p_out, ok_out = canaried(8 * max_out)
            nl, nc = C.c_int64(-7), C.c_int64(-7)
            maddr = addr + mo.value
            got = lib.gmr_bvh_parse_motion(C.c_void_p(maddr), len(data) - mo.value, int(rng.choice([0, 1, 3, 1 << 40])), C.c_void_p(p_out), max_out, C.byref(nl), C.byref(nc))
            good = good and ok_out() and got <= max_out and got >= -3
            rows_ok += got > 0
        if not good:
            bad += 1
            print(f"VIOLATION rc={rc} maxj={maxj} ncap={ncap} len={len(data)} head={data[:80]!r}", flush=True)
    print(f"bvh text fuzz done: {runs} header calls ({parsed} accepted, {rows_ok} with motion values parsed), {bad} violations, {time.time() - t0:.0f} s; "
          f"text ended at a PROT_NONE page, outputs between canaries")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
