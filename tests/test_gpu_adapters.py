"""The two input-adapter kernels (rows f-1, f-2) beyond the reference-generated goldens of test_gpu_api.py: random skeletons of
every lane layout (several frames per wavefront, one frame, two and three joints per lane), frame counts around the run and group
edges, every row layout, column selections, the trigonometric arm of the slerp -- against numpy restatements of the formulas the
kernels cite (Euler -> quaternion, hierarchy-order FK, Z-up turn; slerp / lerp / orientation chaining)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from gmr_amd import _native  # noqa: E402

vp = C.c_void_p


def _qmul(a, b):  # wxyz Hamilton product, broadcasting
    w1, x1, y1, z1 = np.moveaxis(a, -1, 0)
    w2, x2, y2, z2 = np.moveaxis(b, -1, 0)
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=-1)


def _qrot(q, v):  # v + 2 w (u x v) + 2 u x (u x v)
    u, w = q[..., 1:], q[..., :1]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def _axis_quat(ang, axis):
    q = np.zeros(ang.shape + (4,))
    q[..., 0] = np.cos(0.5 * ang)
    q[..., 1 + axis] = np.sin(0.5 * ang)
    return q


def _bvh_restatement(parents, order, lpos, eul_rad, extra_pos, extra_rot, scale):
    """lafan1.py:8-40 + lafan_vendor/utils.py:56-103 in numpy: Euler -> quaternion, FK in hierarchy order, Z-up, scale, extras."""
    lq = _qmul(_axis_quat(eul_rad[..., 0], order[0]), _qmul(_axis_quat(eul_rad[..., 1], order[1]), _axis_quat(eul_rad[..., 2], order[2])))
    T, J = lpos.shape[:2]
    gq, gp = np.zeros((T, J, 4)), np.zeros((T, J, 3))
    gq[:, 0], gp[:, 0] = lq[:, 0], lpos[:, 0]
    for j in range(1, J):
        p = parents[j]
        gp[:, j] = _qrot(gq[:, p], lpos[:, j]) + gp[:, p]
        gq[:, j] = _qmul(gq[:, p], lq[:, j])
    rq = np.array([np.sqrt(0.5), np.sqrt(0.5), 0.0, 0.0])
    gq = _qmul(np.broadcast_to(rq, gq.shape), gq)
    gp = np.stack([gp[..., 0], -gp[..., 2], gp[..., 1]], axis=-1) * scale
    if len(extra_pos):
        gp = np.concatenate([gp, gp[:, extra_pos]], axis=1)
        gq = np.concatenate([gq, gq[:, extra_rot]], axis=1)
    return gp, gq


def _random_tree(rng, J, chain_bias):
    """parents in hierarchy order; chain_bias -> deep chains (many pointer-jumping rounds)."""
    par = np.full(J, -1, dtype=np.int32)
    for j in range(1, J):
        par[j] = j - 1 if rng.random() < chain_bias else rng.integers(0, j)
    return par


def _rows_for(layout, lpos, eul_deg, offsets, scales=None):
    T, J = lpos.shape[:2]
    if layout == 3:
        return np.concatenate([lpos[:, 0], eul_deg.reshape(T, -1)], axis=1)
    if layout == 6:
        return np.concatenate([lpos, eul_deg], axis=2).reshape(T, -1)
    blk = np.concatenate([(lpos[:, 1:] - offsets[None, 1:]) / scales, eul_deg[:, 1:], scales], axis=2)  # 9: offset + position * scale
    return np.concatenate([lpos[:, 0], blk.reshape(T, -1)], axis=1)


def _call_rows(lib, dev, parents, order, extra_pos, extra_rot, layout, offsets, rows, scale, out_cols=None):
    J, E, T = len(parents), len(extra_pos), rows.shape[0]
    B = len(out_cols) if out_cols is not None else J + E
    d_rows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(offsets)).to(dev)
    pos = torch.full((T, B, 3), float("nan"), dtype=torch.float64, device=dev)
    quat = torch.full((T, B, 4), float("nan"), dtype=torch.float64, device=dev)
    par, od = np.ascontiguousarray(parents, np.int32), np.asarray(order, np.int32)
    ep, er = np.asarray(extra_pos, np.int32), np.asarray(extra_rot, np.int32)
    oc = None if out_cols is None else np.asarray(out_cols, np.int32)
    rc = lib.gmr_bvh_fk_rows(par.ctypes.data_as(vp), J, od.ctypes.data_as(vp), ep.ctypes.data_as(vp) if E else None, er.ctypes.data_as(vp) if E else None, E,
                             layout, vp(d_off.data_ptr()), vp(d_rows.data_ptr()), rows.shape[1], T, scale,
                             oc.ctypes.data_as(vp) if oc is not None else None, B, vp(pos.data_ptr()), vp(quat.data_ptr()), None)
    torch.cuda.synchronize()
    return rc, pos.cpu().numpy(), quat.cpu().numpy()


@pytest.mark.parametrize("J,T,layout,bias", [(1, 7, 3, 0.5), (5, 130, 3, 0.9), (16, 257, 6, 0.3), (22, 1001, 3, 0.6), (31, 64, 9, 0.8), (32, 65, 3, 1.0),
                                             (33, 200, 3, 0.5), (64, 129, 6, 1.0), (65, 130, 3, 0.7), (101, 250, 3, 0.6), (128, 33, 9, 0.95),
                                             (129, 40, 3, 0.5), (192, 77, 6, 0.85)])
def test_bvh_fk_rows_random_skeletons(J, T, layout, bias):
    lib = _native.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(J * 1000 + T)
    parents = _random_tree(rng, J, bias)
    order = tuple(int(x) for x in rng.permutation(3))
    offsets = rng.normal(0, 20.0, (J, 3))
    lpos = np.repeat(offsets[None], T, axis=0)
    lpos[:, 0] = np.cumsum(rng.normal(0, 2.0, (T, 3)), axis=0) + [0, 90, 0]
    if layout != 3:
        lpos[:, 1:] += rng.normal(0, 1.0, (T, J - 1, 3)) if J > 1 else 0.0
    eul = rng.uniform(-180.0, 180.0, (T, J, 3)) * (rng.random((1, J, 1)) < 0.8) + rng.normal(0, 400.0, (T, J, 3)) * (rng.random((T, J, 1)) < 0.02)
    scales = rng.uniform(0.5, 2.0, (T, max(J - 1, 0), 3))
    if layout == 9:
        if J < 2:
            pytest.skip("9-channel rows need a non-root joint")
        eul[:, 0] = 0.0   # the layout carries no root rotation
    rows = _rows_for(layout, lpos, eul, offsets, scales)
    if layout == 9:  # what the kernel reconstructs: offset + position * scale
        blk = rows[:, 3:].reshape(T, J - 1, 9)
        lpos = lpos.copy()
        lpos[:, 1:] = offsets[None, 1:] + blk[:, :, 0:3] * blk[:, :, 6:9]
    E = min(2, J)
    extra_pos = [int(x) for x in rng.integers(0, J, E)]
    extra_rot = [int(x) for x in rng.integers(0, J, E)]
    rc, pos, quat = _call_rows(lib, dev, parents, order, extra_pos, extra_rot, layout, offsets, rows, 0.01)
    assert rc == 0
    p_ref, q_ref = _bvh_restatement(parents, order, lpos, np.radians(eul), extra_pos, extra_rot, 0.01)
    scale_p = max(1.0, np.abs(p_ref).max())
    assert np.abs(pos - p_ref).max() < 1e-11 * scale_p * max(1, J // 8)
    assert np.abs(quat - q_ref).max() < 1e-12 * max(1, J // 4)   # unit quaternions; products of up to J of them
    # a column selection is the same numbers in the requested columns, nothing else written
    B = J + E
    sel = [int(x) for x in rng.permutation(B)[: max(1, min(14, B))]]
    rc, pos_s, quat_s = _call_rows(lib, dev, parents, order, extra_pos, extra_rot, layout, offsets, rows, 0.01, out_cols=sel)
    assert rc == 0 and np.array_equal(pos_s, pos[:, sel]) and np.array_equal(quat_s, quat[:, sel])


def test_bvh_fk_split_arrays_entry_equals_rows_entry():
    """gmr_bvh_fk (ABI <= 3: local positions + radians as two arrays) runs the same kernel as gmr_bvh_fk_rows."""
    lib = _native.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    J, T = 22, 500
    parents = _random_tree(rng, J, 0.6)
    offsets = rng.normal(0, 20.0, (J, 3))
    lpos = np.repeat(offsets[None], T, axis=0)
    lpos[:, 0] = rng.normal(0, 50.0, (T, 3))
    eul = rng.uniform(-180, 180, (T, J, 3))
    rc, pos, quat = _call_rows(lib, dev, parents, (2, 1, 0), [3, 7], [4, 8], 3, offsets, _rows_for(3, lpos, eul, offsets), 0.01)
    assert rc == 0
    d_lp, d_er = torch.from_numpy(lpos).to(dev), torch.from_numpy(np.radians(eul)).to(dev)
    p2 = torch.empty((T, J + 2, 3), dtype=torch.float64, device=dev)
    q2 = torch.empty((T, J + 2, 4), dtype=torch.float64, device=dev)
    od, ep, er = np.array([2, 1, 0], np.int32), np.array([3, 7], np.int32), np.array([4, 8], np.int32)
    assert lib.gmr_bvh_fk(parents.ctypes.data_as(vp), J, od.ctypes.data_as(vp), ep.ctypes.data_as(vp), er.ctypes.data_as(vp), 2, vp(d_lp.data_ptr()),
                          vp(d_er.data_ptr()), T, 0.01, vp(p2.data_ptr()), vp(q2.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(p2.cpu().numpy(), pos) and np.array_equal(q2.cpu().numpy(), quat)
    # argument checks
    bad = np.array([0, 5], np.int32)
    assert lib.gmr_bvh_fk_rows(parents.ctypes.data_as(vp), J, od.ctypes.data_as(vp), None, None, 0, 3, vp(d_lp.data_ptr()), vp(d_lp.data_ptr()), 3 + 3 * J + 1, T, 0.01,
                               None, 0, vp(p2.data_ptr()), vp(q2.data_ptr()), None) == -1                      # wrong row length
    assert lib.gmr_bvh_fk_rows(parents.ctypes.data_as(vp), J, od.ctypes.data_as(vp), None, None, 0, 4, vp(d_lp.data_ptr()), vp(d_lp.data_ptr()), 3 + 3 * J, T, 0.01,
                               None, 0, vp(p2.data_ptr()), vp(q2.data_ptr()), None) == -3                      # unknown layout
    dup = np.array([1, 1], np.int32)
    assert lib.gmr_bvh_fk_rows(parents.ctypes.data_as(vp), J, od.ctypes.data_as(vp), None, None, 0, 3, vp(d_lp.data_ptr()), vp(d_lp.data_ptr()), 3 + 3 * J, T, 0.01,
                               dup.ctypes.data_as(vp), 2, vp(p2.data_ptr()), vp(q2.data_ptr()), None) == -1    # a column named twice
    assert bad is not None


def _smplx_restatement(go, fp, jt, parents, T_out, resample):
    """smpl.py:75-107,127-196 in numpy (scipy for the rotation-vector conversions, as the reference)."""
    from scipy.spatial.transform import Rotation as R
    T, J = fp.shape[:2]
    rv = fp.copy()
    rv[:, 0] = go
    pos = np.zeros((T_out, J, 3))
    quat = np.zeros((T_out, J, 4))
    tt = np.linspace(0, T - 1, T_out) if resample else np.arange(T, dtype=np.float64)
    for k, t in enumerate(tt):
        i1 = int(np.floor(t)); i2 = min(i1 + 1, T - 1); a = t - i1
        q1 = R.from_rotvec(rv[i1]).as_quat()
        if resample:
            q2 = R.from_rotvec(rv[i2]).as_quat()
            dot = np.sum(q1 * q2, axis=1)
            q2 = np.where(dot[:, None] < 0, -q2, q2)
            dot = np.abs(dot)
            th0 = np.arccos(np.minimum(dot, 1.0))
            with np.errstate(divide="ignore", invalid="ignore"):
                s1 = np.where(dot > 0.9995, a, np.sin(th0 * a) / np.sin(th0))
                s0 = np.where(dot > 0.9995, 1 - a, np.cos(th0 * a) - dot * np.sin(th0 * a) / np.sin(th0))
            q = s0[:, None] * q1 + s1[:, None] * q2
            lq = R.from_rotvec(R.from_quat(q).as_rotvec())
        else:
            lq = R.from_rotvec(rv[i1])
        rots = []
        for i in range(J):
            rots.append(lq[i] if i == 0 else rots[parents[i]] * lq[i])
            quat[k, i] = rots[i].as_quat(scalar_first=True)
        pos[k] = jt[i1, :J] + a * (jt[i2, :J] - jt[i1, :J])
    return pos, quat


@pytest.mark.parametrize("J,T,skip,big", [(55, 240, 4, False), (55, 241, 1, False), (55, 97, 2, True), (3, 50, 4, True), (24, 130, 1, False), (17, 129, 3, True),
                                          (32, 64, 2, False), (33, 66, 4, True), (64, 40, 1, True)])
def test_smplx_keypoints_random_trees(J, T, skip, big):
    lib = _native.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(J * 100 + T)
    parents = _random_tree(rng, J, 0.6)
    base = rng.normal(0, 0.5, (1, J, 3))
    fp = base + np.cumsum(rng.normal(0, 0.25 if big else 0.02, (T, J, 3)), axis=0)   # big: neighbouring frames far apart -> the trigonometric slerp arm
    if J > 2:
        fp[:, 2] = 0.0           # small-angle series of from_rotvec
        fp[:, 1] *= 4.0          # rotation angles beyond pi: negative-w quaternions
    go = fp[:, 0].copy()
    S = J + 5
    jt = np.cumsum(rng.normal(0, 0.01, (T, S, 3)), axis=0) + rng.normal(0, 0.5, (1, S, 3))
    resample = skip > 1
    T_out = T // skip if resample else T
    d_go, d_fp, d_jt = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (go, fp, jt))

    def call(cols):
        B = J if cols is None else len(cols)
        pos = torch.full((T_out, B, 3), float("nan"), dtype=torch.float64, device=dev)
        quat = torch.full((T_out, B, 4), float("nan"), dtype=torch.float64, device=dev)
        oc = None if cols is None else np.asarray(cols, np.int32)
        rc = lib.gmr_smplx_keypoints_cols(parents.ctypes.data_as(vp), J, S, vp(d_go.data_ptr()), vp(d_fp.data_ptr()), vp(d_jt.data_ptr()), T, T_out, int(resample),
                                          oc.ctypes.data_as(vp) if oc is not None else None, B, vp(pos.data_ptr()), vp(quat.data_ptr()), None)
        torch.cuda.synchronize()
        return rc, pos.cpu().numpy(), quat.cpu().numpy()

    rc, pos, quat = call(None)
    assert rc == 0
    p_ref, q_ref = _smplx_restatement(go, fp, jt, parents, T_out, resample)
    assert np.abs(pos - p_ref).max() < 1e-12
    d = np.minimum(np.abs(quat - q_ref).max(-1), np.abs(quat + q_ref).max(-1))
    assert d.max() < 1e-10, d.max()   # (the trigonometric arm divides by sin(theta_0) >= 0.03)
    sel = [int(x) for x in rng.permutation(J)[: max(1, min(14, J))]]
    rc, pos_s, quat_s = call(sel)
    assert rc == 0 and np.array_equal(pos_s, pos[:, sel]) and np.array_equal(quat_s, quat[:, sel])


def test_adapter_columns_feed_the_ik_directly(golden_dir):
    """`columns=` of both adapters: the dense [T, 14, 7] the IK config consumes gives the same qpos as the full-width arrays."""
    import os
    from gmr_amd import GeneralMotionRetargeting as GMR
    from gmr_amd.bvh import load_lafan1_file
    from gmr_amd.smplx_adapter import SMPLX_PARENTS, get_smplx_data_offline_fast
    g = GMR(src_human="bvh", tgt_robot="unitree_g1", actual_human_height=1.7)
    path = os.path.join(golden_dir, "bvh_lafan_like.bvh")
    full, sub = load_lafan1_file(path), load_lafan1_file(path, columns=g._cm.slot_names)
    assert sub.body_names == list(g._cm.slot_names) and sub.pos.shape[1] == len(g._cm.slot_names) and abs(sub.human_height - full.human_height) < 1e-12
    assert torch.equal(g.retarget_batch(full.pos, full.quat, full.body_names), g.retarget_batch(sub.pos, sub.quat, sub.body_names))
    with pytest.raises(KeyError):
        load_lafan1_file(path, columns=["Hips", "NoSuchBone"])
    rng = np.random.default_rng(1)
    T = 120
    fp = rng.normal(0, 0.3, (1, 55, 3)) + np.cumsum(rng.normal(0, 0.02, (T, 55, 3)), axis=0)
    jt = np.cumsum(rng.normal(0, 0.01, (T, 127, 3)), axis=0) + rng.normal(0, 0.5, (1, 127, 3))
    g2 = GMR("smplx", "unitree_g1")
    pf, qf, nf, _ = get_smplx_data_offline_fast(fp[:, 0], fp.reshape(T, -1), jt, SMPLX_PARENTS, src_fps=120.0)
    ps, qs, ns, _ = get_smplx_data_offline_fast(fp[:, 0], fp.reshape(T, -1), jt, SMPLX_PARENTS, src_fps=120.0, columns=g2._cm.slot_names)
    assert ns == list(g2._cm.slot_names) and ps.shape == (T // 4, 14, 3)
    assert torch.equal(g2.retarget_batch(pf, qf, nf), g2.retarget_batch(ps, qs, ns))


# ------------------------------------------------------------------ the MOTION block parsed on the device (gmr_bvh_parse_motion_device)
def _device_parse(lib, dev, blob: bytes, segs, n_lines, n_cols, max_slow=4096):
    nf = len(segs)
    text = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
    sb = np.array([a for a, _ in segs], np.int64); se = np.array([b for _, b in segs], np.int64)
    nl = np.asarray(n_lines, np.int64)
    rb = np.concatenate([[0], np.cumsum(nl)])[:-1].astype(np.int64)
    rows = torch.full((int(nl.sum()), n_cols), float("nan"), dtype=torch.float64, device=dev)
    status = np.zeros(nf, np.int32); ntok = np.zeros(nf, np.int64)
    slow = np.zeros((max(max_slow, 1), 3), np.int64); n_slow = C.c_int64(0)
    rc = lib.gmr_bvh_parse_motion_device(vp(text.data_ptr()), len(blob), nf, sb.ctypes.data_as(vp), se.ctypes.data_as(vp), nl.ctypes.data_as(vp), n_cols, rb.ctypes.data_as(vp),
                                         vp(rows.data_ptr()), status.ctypes.data_as(vp), ntok.ctypes.data_as(vp), slow.ctypes.data_as(vp), max_slow, C.byref(n_slow), None)
    return rc, rows.cpu().numpy(), status, ntok, slow[: min(n_slow.value, max_slow)], n_slow.value


def test_device_motion_parser_equals_python_float():
    """Every number the device parser writes equals Python's float() of the same token, bit for bit; tokens it does not decide are
    reported with their place; blank lines, tabs, CR LF, a missing final newline, tokens that straddle lane and chunk edges (rows
    of every length around 64 and 4096 bytes), extra rows behind n_lines are as on the host."""
    lib = _native.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    fmts = ["%.6f", "%.4f", "%d", "%.10f", "%.1f", "%.3e", "%.8E", "%+.5f", "%.15g", "%.17g"]
    for case in range(12):
        n_cols = int(rng.integers(1, 140))
        n_lines = int(rng.integers(1, 400))
        sep = [" ", "  ", "\t", " \t "][case % 4]
        eol = ["\n", "\r\n", " \n"][case % 3]
        vals = rng.normal(0, 1, (n_lines + 3, n_cols)) * 10.0 ** rng.integers(-8, 9, (n_lines + 3, n_cols))
        vals[rng.random(vals.shape) < 0.1] = 0.0
        lines = []
        for r in range(n_lines + 3):
            toks = [(fmts[int(rng.integers(len(fmts)))] % v) for v in vals[r]]
            lines.append(("  " if r % 5 == 0 else "") + sep.join(toks))
            if r % 7 == 3:
                lines.append("   ")  # a blank line
        text = (eol.join(lines) + (eol if case % 2 else "")).encode()
        junk = b"x" * int(rng.integers(0, 40))  # the segment does not start at offset 0
        blob = junk + text + b"   trailing bytes of the next file"
        rc, rows, status, ntok, slow, ns = _device_parse(lib, dev, blob, [(len(junk), len(junk) + len(text))], [n_lines], n_cols)
        assert rc == 0 and status[0] == 0 and ns == len(slow), (case, status, ns)
        exp = np.array([[float(t) for t in ln.split()] for ln in lines if ln.strip()][:n_lines])
        got = rows.copy()
        for k, t, b in slow:   # what the device left to the host: long mantissas (%.17g and friends)
            tok = blob[int(b):].split()[0]
            assert float(tok) == exp.reshape(-1)[int(t)]
            got.reshape(-1)[int(t)] = float(tok)
        assert got.tobytes() == exp.tobytes(), case
        assert ntok[0] == sum(len(ln.split()) for ln in lines)


def test_device_motion_parser_structure_and_slow_path():
    lib = _native.load()
    dev = torch.device("cuda", 0)
    good = b"1 2 3\n4 5 6\n7 8 9\n"
    # several files in one call, each with its own segment; an empty file; a file with fewer rows than asked
    blob = good + b"#" * 5 + b"10 11 12\n" + b"13 14 15"
    segs = [(0, len(good)), (len(good) + 5, len(blob)), (3, 3)]
    rc, rows, status, ntok, slow, ns = _device_parse(lib, dev, blob, segs, [3, 2, 0], 3)
    assert rc == 0 and list(status) == [0, 0, 0] and ns == 0 and np.array_equal(rows, np.arange(1, 16, dtype=np.float64).reshape(5, 3))
    rc, rows, status, *_ = _device_parse(lib, dev, good, [(0, len(good))], [4], 3)
    assert rc == 0 and status[0] == 2                      # fewer rows than the header says
    for bad in (b"1 2 3\n4 5\n6 7 8 9\n", b"1 2\n3 4 5 6\n7 8 9\n", b"1 2 3 4 5 6\n7 8 9\n"):
        rc, rows, status, *_ = _device_parse(lib, dev, bad, [(0, len(bad))], [3 if bad.count(b"\n") == 3 else 2], 3)
        assert rc == 0 and status[0] & 1, bad              # ragged rows: the host parser is asked
    # ... also when only the LAST row asked for is too long (nothing behind it is read, but its extra tokens make the host parser refuse the
    # file; found by tools/experiments/hostile_text_only.py), while extra ROWS behind it are fine
    for blob2, n_lines2, bad2 in ((b"1 2 3\n4 5 6 7\n", 2, True), (b"1 2 3 4\n", 1, True), (b"1 2 3\n4 5 6\n7 8 9\n", 2, False), (b"1 2 3\n4 5 6\n\n  \n7\n", 2, False)):
        rc, rows, status, *_ = _device_parse(lib, dev, blob2, [(0, len(blob2))], [n_lines2], 3)
        assert rc == 0 and bool(status[0] & 1) == bad2, (blob2, status)
    # tokens the device must not decide: reported, in order of their place
    odd = b"1e400 nan 0.1234567890123456789012 12345678901234567890 -inf 1_0 abc 1e 0x10 +.5 5. .e1 1.5e+3 -0 1e-30 9007199254740993\n"
    toks = odd.split()
    rc, rows, status, ntok, slow, ns = _device_parse(lib, dev, odd, [(0, len(odd))], [1], len(toks))
    assert rc == 0 and status[0] == 0
    decided = {int(t) for _, t, _ in slow}
    for i, tk in enumerate(toks):
        if i in decided:
            assert odd[int(slow[[int(t) for _, t, _ in slow].index(i)][2]):].split()[0] == tk   # the byte offset names the token
        else:
            assert rows[0, i] == float(tk) and np.signbit(rows[0, i]) == np.signbit(float(tk)), tk
    assert {toks.index(t) for t in (b"1e400", b"nan", b"-inf", b"1_0", b"abc", b"1e", b"0x10", b".e1", b"0.1234567890123456789012", b"12345678901234567890", b"1e-30", b"9007199254740993")} <= decided
    assert not ({toks.index(t) for t in (b"+.5", b"5.", b"1.5e+3", b"-0")} & decided)
    # a token longer than a lane can follow (past its 32-byte halo) is left to the host too
    long_tok = b"0." + b"0" * 120 + b"1 2\n"
    rc, rows, status, ntok, slow, ns = _device_parse(lib, dev, long_tok, [(0, len(long_tok))], [1], 2)
    assert rc == 0 and ns == 1 and int(slow[0][1]) == 0 and rows[0, 1] == 2.0


def test_bvh_folder_device_parse_equals_host_parse(golden_dir, tmp_path):
    """load_lafan1_files(parse="device") == parse="host" bit for bit on the golden files and on a synthetic folder (several files,
    unequal lengths, a CR LF file, one number written with 25 digits); a ragged file raises the host parser's error; batches read
    ahead (iter_lafan1_batches) give the same clips."""
    import os
    import shutil
    from gmr_amd.bvh import iter_lafan1_batches, load_lafan1_files
    sys_tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import sys
    sys.path.insert(0, sys_tools)
    from config3_files_bench import write_files
    for name in ("bvh_canonical_40f", "bvh_lafan_like", "bvh_pruned_mid_24f", "bvh_nine_channel"):
        p = os.path.join(golden_dir, name + ".bvh")
        st = {}
        d, h = load_lafan1_files([p, p], parse="device", stats=st), load_lafan1_files([p, p], parse="host")
        assert torch.equal(d.pos, h.pos) and torch.equal(d.quat, h.quat) and d.human_heights == h.human_heights and d.body_names == h.body_names
        assert np.array_equal(d.seq_offsets, h.seq_offsets) and st["files_reparsed_on_host"] == 0
    files = write_files(str(tmp_path), 5, 300)
    raw = open(files[1], "rb").read()
    open(files[1], "wb").write(raw.replace(b"\n", b"\r\n"))
    txt = open(files[2], "rb").read().split(b"\n")
    row = txt[-2].split()
    row[7] = b"12.3456789012345678901234"
    txt[-2] = b" ".join(row)
    open(files[2], "wb").write(b"\n".join(txt[:-2]) + b"\n" + txt[-2] + b"\n")
    st = {}
    d, h = load_lafan1_files(files, parse="device", stats=st), load_lafan1_files(files, parse="host")
    assert torch.equal(d.pos, h.pos) and torch.equal(d.quat, h.quat) and d.human_heights == h.human_heights and st["slow_tokens"] == 1
    got = list(iter_lafan1_batches(files, batch_files=2))
    assert [len(b) for b in got] == [2, 2, 1] and torch.equal(torch.cat([b.pos for b in got]), h.pos)
    bad = str(tmp_path / "ragged.bvh")
    shutil.copy(files[0], bad)
    t = open(bad, "rb").read().split(b"\n")
    t[-5] = b" ".join(t[-5].split()[:-1])
    open(bad, "wb").write(b"\n".join(t))
    with pytest.raises(ValueError, match="malformed motion block"):
        load_lafan1_files([files[0], bad], parse="device")


def test_folder_to_pickles_pipeline_of_integration_md(golden_dir, tmp_path):
    """INTEGRATION.md section 1, the folder loop of scripts/bvh_to_robot_dataset.py:59-151 on this engine: batches read ahead and parsed on
    the device, only the columns the config reads, verified chunks, files written by the pool while the next batch solves; the pickles
    are the serial path's, byte for byte, and read back through the reference's reader contract."""
    import os
    import pickle
    import shutil
    from gmr_amd import GeneralMotionRetargeting as GMR, dataset
    from gmr_amd.bvh import iter_lafan1_batches, load_lafan1_file
    src = os.path.join(golden_dir, "bvh_lafan_like.bvh")
    files = []
    for i in range(5):
        files.append(str(tmp_path / f"clip{i}.bvh"))
        shutil.copy(src, files[-1])
    g = GMR(src_human="bvh", tgt_robot="unitree_g1")
    assert g.ik_columns == list(g._cm.slot_names) and len(g.ik_columns) == 14
    outs = [str(tmp_path / "out" / f"clip{i}.pkl") for i in range(5)]
    k = 0
    with dataset.MotionWriter(workers=3) as w:
        for batch in iter_lafan1_batches(files, batch_files=2, columns=g.ik_columns):
            motions = dataset.retarget_clips(g, batch.pos, batch.quat, batch.body_names, batch.seq_offsets, fps=30, height_adjust=False,
                                             root_origin_offset=False, chunk=8, burn_in=8, human_heights=batch.human_heights)
            w.submit(motions, outs[k:k + len(batch)])
            k += len(batch)
    assert w.written == 5 and k == 5
    # the serial path on one file: full-width key-points, no chunks, stock pickle
    clip = load_lafan1_file(src)
    ref = dataset.retarget_clips(GMR(src_human="bvh", tgt_robot="unitree_g1"), clip.pos, clip.quat, clip.body_names, [0, len(clip)], fps=30,
                                 height_adjust=False, root_origin_offset=False, human_heights=[clip.human_height])[0]
    for o in outs:
        d, fps, rp, rr_wxyz, dp, lb, names = dataset.load_robot_motion(o)
        assert fps == 30 and names == g.model.body_names and np.abs(dp - ref["dof_pos"]).max() < 1e-6 and np.abs(rp - ref["root_pos"]).max() < 1e-6
        dataset.validate_motion(d, nq=36)
        assert open(o, "rb").read() == pickle.dumps(d)


def test_bvh_folder_skip_errors(golden_dir, tmp_path):
    """skip_errors: the per-file try / except of scripts/bvh_to_robot_dataset.py:75-80 -- a missing file, a file that is not BVH, one of
    another skeleton and one with a ragged motion row are left out and reported; the good clips are what they are without them."""
    import os
    import shutil
    from gmr_amd.bvh import iter_lafan1_batches, load_lafan1_files
    good = os.path.join(golden_dir, "bvh_lafan_like.bvh")
    other = os.path.join(golden_dir, "bvh_pruned_mid_24f.bvh")
    junk = str(tmp_path / "junk.bvh")
    open(junk, "w").write("this is not a BVH file\n")
    ragged = str(tmp_path / "ragged.bvh")
    t = open(good, "rb").read().split(b"\n")
    t[-4] = b" ".join(t[-4].split()[:-2])
    open(ragged, "wb").write(b"\n".join(t))
    g2 = str(tmp_path / "again.bvh")
    shutil.copy(good, g2)
    files = [good, str(tmp_path / "missing.bvh"), junk, other, ragged, g2]
    with pytest.raises((ValueError, OSError)):
        load_lafan1_files(files)
    b = load_lafan1_files(files, skip_errors=True)
    ref = load_lafan1_files([good, g2])
    assert b.files == [good, g2] and sorted(f for f, _ in b.skipped) == sorted([files[1], junk, other, ragged]) and all(r for _, r in b.skipped)
    assert torch.equal(b.pos, ref.pos) and torch.equal(b.quat, ref.quat) and np.array_equal(b.seq_offsets, ref.seq_offsets) and b.human_heights == ref.human_heights
    # (a batch takes its skeleton from its first good file)
    got = list(iter_lafan1_batches([good, files[1], junk, g2, other, ragged], batch_files=3, skip_errors=True))
    assert [bb.files for bb in got] == [[good], [g2]] and sum(len(bb.skipped) for bb in got) == 4
    only_bad = list(iter_lafan1_batches([junk, ragged], batch_files=1, skip_errors=True))
    assert [len(bb) for bb in only_bad] == [0, 0] and len(only_bad[0].skipped) == 1


def test_smplx_joint_files_equal_per_clip_adapter_calls(tmp_path):
    """smplx_adapter.load_joint_files / iter_joint_batches (the file side of row f-2): every clip of the batch is, bit for bit, what
    get_smplx_data_offline_fast returns for that clip's arrays -- 120 -> 30 fps and 1:1 clips in one batch, float32 files (a body model's
    dtype) and float64 ones, all 55 columns and a config's 14; heights follow load_smplx_file's betas rule (utils/smpl.py:37-40), fps
    the aligned rate (:172); a broken file is reported, not fatal, with skip_errors."""
    from gmr_amd import GeneralMotionRetargeting as GMR, synth
    from gmr_amd import smplx_adapter as sa
    dev = torch.device("cuda", 0)
    g = GMR(src_human="smplx", tgt_robot="unitree_g1")
    files, arrays = [], []
    for k, (T, fps, dt) in enumerate([(481, 120.0, np.float32), (90, 30.0, np.float64), (7, 120.0, np.float32), (250, 60.0, np.float64)]):
        go, fp, jt = (a.cpu().numpy().astype(dt) for a in synth.smplx_arrays_torch(T, dev, seed=k))
        betas = np.zeros(16) if k % 2 else np.zeros((1, 16))
        betas.reshape(-1)[0] = 0.3 * k - 0.4
        f = str(tmp_path / f"c{k}.npz")
        sa.save_joint_file(f, jt, go, fp.reshape(T, -1), fps, betas)
        files.append(f)
        arrays.append((go, fp, jt[:, :55], fps, 1.66 + 0.1 * (0.3 * k - 0.4)))
    z = np.load(files[0])
    assert z["joints"].shape == (481, 55, 3) and z["joints"].dtype == np.float32 and z["full_pose"].shape == (481, 165)
    for cols in (None, g.ik_columns):
        b = sa.load_joint_files(files, columns=cols)
        assert len(b) == 4 and b.files == files and b.body_names == (sa.SMPLX_JOINT_NAMES if cols is None else cols)
        assert b.seq_offsets.tolist() == [0, 120, 210, 211, 336]
        for k, (go, fp, jt, fps, h) in enumerate(arrays):
            p, q, names, afps = sa.get_smplx_data_offline_fast(go.astype(np.float64), fp.astype(np.float64), jt.astype(np.float64), src_fps=fps, columns=cols)
            a, e = b.seq_offsets[k], b.seq_offsets[k + 1]
            assert torch.equal(b.pos[a:e], p) and torch.equal(b.quat[a:e], q) and b.fps[k] == afps
            assert abs(b.human_heights[k] - h) < 1e-12
    bad = str(tmp_path / "broken.npz")
    np.savez(bad, joints=np.zeros((4, 10, 3)), global_orient=np.zeros((4, 3)), full_pose=np.zeros((4, 165)), mocap_frame_rate=30.0, betas=np.zeros(16))
    with pytest.raises(ValueError):
        sa.load_joint_files(files[:1] + [bad])
    got = list(sa.iter_joint_batches([files[0], bad, str(tmp_path / "missing.npz"), files[1]], batch_files=2, skip_errors=True))
    assert [bb.files for bb in got] == [[files[0]], [files[1]]] and [len(bb.skipped) for bb in got] == [1, 1]
    assert torch.equal(got[1].pos, sa.load_joint_files([files[1]]).pos)


def test_smplx_joint_files_to_pickles_pipeline(tmp_path):
    """The loop body of scripts/smplx_to_robot_dataset.py:63-146 behind the body model: joint-array files -> adapter -> batched IK with one
    actual_human_height per file -> FK / height adjust / xy offset -> pickles written by the pool.  The files hold robot-consistent
    key-points (gmr_amd.synth.write_smplx_joint_files), so the result must be what solving those key-points in memory gives."""
    from gmr_amd import GeneralMotionRetargeting as GMR, dataset, synth
    from gmr_amd import smplx_adapter as sa
    dev = torch.device("cuda", 0)
    g = GMR(src_human="smplx", tgt_robot="unitree_g1")
    cm = g._cm
    lens = np.array([120, 77, 200, 64, 150])
    heights = [1.6, 1.7, 1.75, 1.8, 1.66]
    pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=3, device=dev, hard=np.arange(5) % 2 == 1, yaw0=1.0, dtype=torch.float64)
    files = synth.write_smplx_joint_files(str(tmp_path), pos, quat, names, offs, fps=30.0, heights=heights, dtype=np.float64)
    outs = [str(tmp_path / "out" / f"m{i}.pkl") for i in range(5)]
    k, kp_err = 0, 0.0
    with dataset.MotionWriter(workers=2) as w:
        for batch in sa.iter_joint_batches(files, batch_files=2, columns=g.ik_columns):
            a, e = offs[k], offs[k + len(batch)]
            cols = [names.index(c) for c in batch.body_names]
            kp_err = max(kp_err, float((batch.pos - pos[a:e][:, cols]).abs().max()), float(torch.minimum((batch.quat - quat[a:e][:, cols]).abs().amax(-1), (batch.quat + quat[a:e][:, cols]).abs().amax(-1)).max()))
            assert np.allclose(batch.human_heights, heights[k:k + len(batch)], atol=1e-12) and batch.fps == [30.0] * len(batch)
            motions = dataset.retarget_clips(g, batch.pos, batch.quat, batch.body_names, batch.seq_offsets, fps=batch.fps, human_heights=batch.human_heights)
            w.submit(motions, outs[k:k + len(batch)])
            k += len(batch)
    assert w.written == 5 and kp_err < 1e-9
    ref = dataset.retarget_clips(GMR(src_human="smplx", tgt_robot="unitree_g1"), pos, quat, names, offs, fps=30.0, human_heights=heights)
    for o, r in zip(outs, ref):
        d, fps, rp, rr, dp, lb, bn = dataset.load_robot_motion(o)
        dataset.validate_motion(d, nq=36)
        assert fps == 30.0 and np.abs(dp - r["dof_pos"]).max() < 1e-6 and np.abs(rp - r["root_pos"]).max() < 1e-6


def test_dataset_script_twins_end_to_end(golden_dir, tmp_path, capsys):
    """python -m gmr_amd.scripts.{bvh,smplx}_to_robot_dataset: the reference's flags, folder walk, exclusions and output files
    (scripts/bvh_to_robot_dataset.py:59-151, scripts/smplx_to_robot_dataset.py:171-245) with the loops run on the GPU; a second run skips what
    exists, --override rewrites it, a broken file is reported and skipped, another skeleton in the same folder is converted in its own batch."""
    import os
    import shutil
    from gmr_amd import dataset, synth, GeneralMotionRetargeting as GMR
    from gmr_amd.scripts import bvh_to_robot_dataset, smplx_to_robot_dataset
    src, tgt = str(tmp_path / "bvh_in"), str(tmp_path / "bvh_out")
    os.makedirs(os.path.join(src, "sub"))
    for n in ("a1.bvh", "a2.bvh", os.path.join("sub", "b1.bvh")):
        shutil.copy(os.path.join(golden_dir, "bvh_lafan_like.bvh"), os.path.join(src, n))
    # the same bones with another limb length: a second skeleton in the folder; and a skeleton without the bones bvh_to_g1.json names
    txt = open(os.path.join(golden_dir, "bvh_lafan_like.bvh")).read().split("\n")
    k = [i for i, ln in enumerate(txt) if "OFFSET" in ln][3]
    txt[k] = txt[k].replace("OFFSET", "OFFSET 0.5 0.25 0.125 #").split("#")[0]
    open(os.path.join(src, "other_skeleton.bvh"), "w").write("\n".join(txt))
    shutil.copy(os.path.join(golden_dir, "bvh_canonical_40f.bvh"), os.path.join(src, "zz_foreign_bones.bvh"))
    open(os.path.join(src, "broken.bvh"), "w").write("HIERARCHY\nROOT x {")
    open(os.path.join(src, "readme.txt"), "w").write("not motion")
    assert bvh_to_robot_dataset.main(["--src_folder", src, "--tgt_folder", tgt, "--robot", "unitree_g1", "--batch_files", "8"]) == 0
    out = capsys.readouterr().out
    assert "Error loading" in out and "broken.bvh" in out and "zz_foreign_bones.bvh" in out and "no such bone" in out and "Done." in out
    made = sorted(os.path.relpath(os.path.join(d, f), tgt) for d, _, fs in os.walk(tgt) for f in fs)
    assert made == ["a1.pkl", "a2.pkl", "other_skeleton.pkl", os.path.join("sub", "b1.pkl")]
    for m in made:
        d, fps, rp, rr, dp, lb, names = dataset.load_robot_motion(os.path.join(tgt, m))
        dataset.validate_motion(d, nq=36)
        assert fps == 30
    t0 = os.path.getmtime(os.path.join(tgt, "a1.pkl"))
    assert bvh_to_robot_dataset.main(["--src_folder", src, "--tgt_folder", tgt]) == 0
    assert "(4 skipped: target exists)" in capsys.readouterr().out and os.path.getmtime(os.path.join(tgt, "a1.pkl")) == t0
    # SMPL-X joint files
    dev = torch.device("cuda", 0)
    g = GMR(src_human="smplx", tgt_robot="unitree_g1")
    lens = np.array([60, 45, 30, 20])
    pos, quat, names, offs = synth.synth_clips_torch(g._cm, lens, seed=8, device=dev, yaw0=1.0, dtype=torch.float64)
    s2, t2 = str(tmp_path / "sm_in"), str(tmp_path / "sm_out")
    os.makedirs(s2)
    files = synth.write_smplx_joint_files(s2, pos, quat, names, offs, fps=30.0, heights=[1.7, 1.6, 1.8, 1.75])
    os.rename(files[2], os.path.join(s2, "clip_crawl_7.npz"))          # excluded by name (:218-227)
    os.rename(files[3], os.path.join(s2, "subject_stagei.npz"))        # excluded by suffix (:208-209)
    assert smplx_to_robot_dataset.main(["--src_folder", s2, "--tgt_folder", t2, "--robot", "unitree_g1", "--num_cpus", "2", "--hard_motions"]) == 0
    out = capsys.readouterr().out
    assert "full args_list: 3" in out and "new args_list: 2" in out
    assert sorted(os.listdir(t2)) == ["clip_00000.pkl", "clip_00001.pkl"]
    ref = dataset.retarget_clips(GMR(src_human="smplx", tgt_robot="unitree_g1"), pos[: offs[2]], quat[: offs[2]], names, offs[:3], fps=30.0, human_heights=[1.7, 1.6])
    for k in range(2):
        d, fps, rp, rr, dp, lb, bn = dataset.load_robot_motion(os.path.join(t2, f"clip_{k:05d}.pkl"))
        dataset.validate_motion(d, nq=36)
        assert fps == 30.0 and np.abs(dp - ref[k]["dof_pos"]).max() < 1e-4 and np.abs(rp - ref[k]["root_pos"]).max() < 1e-4


def test_utils_modules_keep_the_reference_signatures(golden_dir):
    """gmr_amd.utils.lafan1.load_lafan1_file / gmr_amd.utils.smpl.get_smplx_data_offline_fast: the import paths, arguments and return shapes of
    general_motion_retargeting.utils (lafan1.py:8-71, smpl.py:109-198) -- a list of per-frame dicts {name: (position, quaternion wxyz)} plus the
    height / the aligned frame rate -- so that a script keeps its loops; values against the reference loader's golden and the scipy restatement."""
    import os
    from types import SimpleNamespace
    from gmr_amd.smplx_adapter import SMPLX_JOINT_NAMES, SMPLX_PARENTS
    from gmr_amd.utils.lafan1 import load_lafan1_file
    from gmr_amd.utils.smpl import get_smplx_data_offline_fast, load_smplx_file
    g = np.load(os.path.join(golden_dir, "bvh_lafan_like.npz"))
    frames, height = load_lafan1_file(os.path.join(golden_dir, "bvh_lafan_like.bvh"))
    assert isinstance(frames, list) and len(frames) == g["pos"].shape[0] and list(frames[0].keys()) == [str(n) for n in g["names"]]
    assert abs(height - float(g["human_height"])) < 1e-9
    for t in (0, len(frames) - 1):
        for i, n in enumerate(frames[t]):
            p, q = frames[t][n]
            assert p.shape == (3,) and q.shape == (4,) and np.abs(p - g["pos"][t, i]).max() < 1e-9
            assert min(np.abs(q - g["quat"][t, i]).max(), np.abs(q + g["quat"][t, i]).max()) < 1e-9
    # SMPL-X: what load_smplx_file hands over, faked (float32 torch tensors as the body model emits them)
    rng = np.random.default_rng(4)
    for T, fps, tgt in ((120, 120.0, 30), (40, 30.0, 30)):
        fp = (rng.normal(0, 0.4, (1, 55, 3)) + np.cumsum(rng.normal(0, 0.02, (T, 55, 3)), axis=0)).astype(np.float32)
        jt = (rng.normal(0, 0.5, (1, 127, 3)) + np.cumsum(rng.normal(0, 0.01, (T, 127, 3)), axis=0)).astype(np.float32)
        data = {"mocap_frame_rate": np.array(fps), "pose_body": np.zeros((T, 63))}
        model = SimpleNamespace(parents=torch.tensor(SMPLX_PARENTS))
        out = SimpleNamespace(global_orient=torch.from_numpy(fp[:, 0].copy()), full_pose=torch.from_numpy(fp.reshape(T, 165)), joints=torch.from_numpy(jt))
        fr, afps = get_smplx_data_offline_fast(data, model, out, tgt_fps=tgt)
        resample = tgt < fps
        T_out = T // int(fps / tgt) if resample else T
        assert len(fr) == T_out and list(fr[0].keys()) == SMPLX_JOINT_NAMES
        assert (afps == T_out / T * fps) if resample else (afps is tgt)
        p_ref, q_ref = _smplx_restatement(fp[:, 0].astype(np.float64), fp.astype(np.float64), jt.astype(np.float64), np.asarray(SMPLX_PARENTS, np.int32), T_out, resample)
        for t in (0, T_out - 1):
            for i, n in enumerate(SMPLX_JOINT_NAMES):
                p, q = fr[t][n]
                assert np.abs(p - p_ref[t, i]).max() < 1e-12 and min(np.abs(q - q_ref[t, i]).max(), np.abs(q + q_ref[t, i]).max()) < 1e-10
    with pytest.raises(ImportError, match="smplx"):
        load_smplx_file("nothing.npz", "nowhere")
