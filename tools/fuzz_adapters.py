#!/usr/bin/env python3
"""Randomised differential run of the input adapters (rows f-1, f-2) and the device text parser against their restatements
(tests/test_gpu_adapters.py: numpy FK / slerp restatements of the cited reference lines, Python float() for the parser).

    python tools/fuzz_adapters.py [seconds] [seed]

Random skeletons (1-192 joints: every lane layout), all three BVH row layouts, Euler orders with repeated axes, frame counts around the
batch / run edges, column selections; SMPL-X trees of 2-64 joints with and without resampling, small and large frame-to-frame rotations;
MOTION text with random number formats, separators, line endings, blank lines and segment offsets.  Prints one summary line per family.
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gmr_amd import _native
import tests.test_gpu_adapters as T


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    lib = _native.load()
    dev = torch.device("cuda", 0)
    t_end = time.time() + budget
    stats = {"bvh": [0, 0, 0.0, 0.0], "smplx": [0, 0, 0.0, 0.0], "text": [0, 0, 0, 0]}
    while time.time() < t_end:
        # ---- BVH
        J = int(rng.choice([rng.integers(1, 33), rng.integers(33, 65), rng.integers(65, 193)], p=[0.6, 0.25, 0.15]))
        layout = int(rng.choice([3, 6, 9])) if J > 1 else int(rng.choice([3, 6]))
        Tn = int(rng.integers(1, 700))
        parents = T._random_tree(rng, J, float(rng.random()))
        order = tuple(int(x) for x in (rng.permutation(3) if rng.random() < 0.85 else rng.integers(0, 3, 3)))
        offsets = rng.normal(0, 20.0, (J, 3))
        lpos = np.repeat(offsets[None], Tn, axis=0)
        lpos[:, 0] = rng.normal(0, 100.0, (Tn, 3))
        if layout != 3 and J > 1:
            lpos[:, 1:] += rng.normal(0, 1.0, (Tn, J - 1, 3))
        eul = rng.uniform(-180.0, 180.0, (Tn, J, 3)) if rng.random() < 0.8 else rng.normal(0, 500.0, (Tn, J, 3))
        scales = rng.uniform(0.5, 2.0, (Tn, max(J - 1, 0), 3))
        if layout == 9:
            eul[:, 0] = 0.0
        rows = T._rows_for(layout, lpos, eul, offsets, scales)
        if layout == 9:
            blk = rows[:, 3:].reshape(Tn, J - 1, 9)
            lpos = lpos.copy()
            lpos[:, 1:] = offsets[None, 1:] + blk[:, :, 0:3] * blk[:, :, 6:9]
        E = int(rng.integers(0, min(3, J) + 1))
        ep = [int(x) for x in rng.permutation(J)[:E]]
        er = [int(x) for x in rng.permutation(J)[:E]]
        rc, pos, quat = T._call_rows(lib, dev, parents, order, ep, er, layout, offsets, rows, 0.01)
        assert rc == 0
        p_ref, q_ref = T._bvh_restatement(parents, order, lpos, np.radians(eul), ep, er, 0.01)
        s = stats["bvh"]
        s[0] += 1; s[1] += Tn
        s[2] = max(s[2], float(np.abs(pos - p_ref).max() / max(1.0, np.abs(p_ref).max())))
        s[3] = max(s[3], float(np.abs(quat - q_ref).max()))
        B = J + E
        sel = [int(x) for x in rng.permutation(B)[: int(rng.integers(1, B + 1))]]
        rc, pos_s, quat_s = T._call_rows(lib, dev, parents, order, ep, er, layout, offsets, rows, 0.01, out_cols=sel)
        assert rc == 0 and np.array_equal(pos_s, pos[:, sel]) and np.array_equal(quat_s, quat[:, sel])
        # ---- SMPL-X
        J = int(rng.integers(2, 65)); Tn = int(rng.integers(2, 260)); skip = int(rng.choice([1, 2, 3, 4]))
        parents = T._random_tree(rng, J, float(rng.random()))
        fp = rng.normal(0, 0.6, (1, J, 3)) + np.cumsum(rng.normal(0, float(rng.choice([0.01, 0.05, 0.3])), (Tn, J, 3)), axis=0)
        S = J + int(rng.integers(0, 70))
        jt = rng.normal(0, 1.0, (Tn, S, 3))
        resample = skip > 1 and Tn // skip >= 1
        T_out = Tn // skip if resample else Tn
        d_go, d_fp, d_jt = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (fp[:, 0].copy(), fp, jt))
        pos = torch.full((T_out, J, 3), float("nan"), dtype=torch.float64, device=dev)
        quat = torch.full((T_out, J, 4), float("nan"), dtype=torch.float64, device=dev)
        vp = T.vp
        rc = lib.gmr_smplx_keypoints_cols(parents.ctypes.data_as(vp), J, S, vp(d_go.data_ptr()), vp(d_fp.data_ptr()), vp(d_jt.data_ptr()), Tn, T_out, int(resample),
                                          None, J, vp(pos.data_ptr()), vp(quat.data_ptr()), None)
        torch.cuda.synchronize()
        assert rc == 0
        p_ref, q_ref = T._smplx_restatement(fp[:, 0].copy(), fp, jt, parents, T_out, resample)
        q = quat.cpu().numpy()
        s = stats["smplx"]
        s[0] += 1; s[1] += T_out
        s[2] = max(s[2], float(np.abs(pos.cpu().numpy() - p_ref).max()))
        s[3] = max(s[3], float(np.minimum(np.abs(q - q_ref).max(-1), np.abs(q + q_ref).max(-1)).max()))
        # ---- text
        n_cols = int(rng.integers(1, 200)); n_lines = int(rng.integers(1, 300))
        fmts = ["%.6f", "%.4f", "%d", "%.10f", "%.1f", "%.3e", "%.8E", "%+.5f", "%.15g", "%.17g", "%.0f."]
        sep = [" ", "  ", "\t", " \t "][int(rng.integers(4))]
        eol = ["\n", "\r\n", " \n"][int(rng.integers(3))]
        vals = rng.normal(0, 1, (n_lines, n_cols)) * 10.0 ** rng.integers(-10, 11, (n_lines, n_cols))
        lines = []
        for r in range(n_lines):
            lines.append(sep.join(fmts[int(rng.integers(len(fmts)))] % v for v in vals[r]))
            if rng.random() < 0.1:
                lines.append(" ")
        text = (eol.join(lines) + (eol if rng.random() < 0.5 else "")).encode()
        junk = b"#" * int(rng.integers(0, 100))
        blob = junk + text + b" tail"
        rc, rows_d, status, ntok, slow, ns = T._device_parse(lib, dev, blob, [(len(junk), len(junk) + len(text))], [n_lines], n_cols, max_slow=1 << 16)
        assert rc == 0 and status[0] == 0, (status, n_cols, n_lines)
        exp = np.array([[float(t) for t in ln.split()] for ln in lines if ln.strip()])
        got = rows_d.copy()
        for k, t, b in slow:
            got.reshape(-1)[int(t)] = float(blob[int(b):].split()[0])
        assert got.tobytes() == exp.tobytes()
        s = stats["text"]
        s[0] += 1; s[1] += n_lines * n_cols; s[2] += int(ns); s[3] += len(text)
        # ---- hostile text: mutated / random bytes.  The device parser must never guess: whenever the host parser accepts the block
        # (n_lines rows of n_cols numbers), the device's values, with its reported tokens patched by float(), are the host's bit for bit;
        # whatever the host rejects the device flags (status) or hands over (slow list) -- and nothing faults.
        h = stats.setdefault("hostile", [0, 0, 0, 0])
        for _ in range(4):
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
            if not text:
                continue
            junk = b"#" * int(rng.integers(0, 70))
            blob = junk + text
            rc, rows_d, status, ntok, slow, ns = T._device_parse(lib, dev, blob, [(len(junk), len(junk) + len(text))], [n_lines], n_cols, max_slow=1 << 14)
            assert rc == 0, rc
            hout = np.full(n_lines * n_cols + 8, np.nan)
            nl_h, nc_h = C.c_int64(0), C.c_int64(0)
            got_h = lib.gmr_bvh_parse_motion(text, len(text), n_lines, hout.ctypes.data_as(vp), n_lines * n_cols, C.byref(nl_h), C.byref(nc_h))
            h[0] += 1
            if got_h == n_lines * n_cols and nc_h.value == n_cols:
                h[1] += 1
                assert status[0] == 0 and ns == len(slow), (status, ns, text[:80])
                got = rows_d.copy().reshape(-1)
                for k, t, b in slow:
                    got[int(t)] = float(blob[int(b):].split()[0])
                assert got.tobytes() == hout[: n_lines * n_cols].tobytes(), text[:120]
            else:
                h[2] += int(status[0] != 0 or ns > 0 or int(ntok[0]) < n_lines * n_cols)
                h[3] += 1
    b, sx, tx = stats["bvh"], stats["smplx"], stats["text"]
    print(f"seed {seed}, {budget:.0f} s")
    print(f"bvh_fk_kernel:           {b[0]} batches, {b[1]} frames: worst |pos - restatement| (relative) {b[2]:.2e}, worst |quat - restatement| {b[3]:.2e}; column selections bitwise equal")
    print(f"smplx_keypoints_kernel:  {sx[0]} batches, {sx[1]} output frames: worst |pos| {sx[2]:.2e}, worst |quat| (up to sign) {sx[3]:.2e}")
    hs = stats.get("hostile", [0, 0, 0, 0])
    print(f"hostile text:            {hs[0]} blocks of truncated / flipped / random bytes: {hs[1]} accepted by the host parser -> device values identical; "
          f"{hs[3]} rejected by the host parser -> {hs[2]} of them flagged or handed over by the device")
    print(f"bvh_txt_parse_kernel:    {tx[0]} blocks, {tx[1]} numbers, {tx[3]} bytes: every number == float(token) bit for bit ({tx[2]} handed to the host as off the exact path)")


if __name__ == "__main__":
    main()
