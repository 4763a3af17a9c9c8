"""MJCF kinematic-subset compiler.

Reads only what the retarget hot path needs from a MuJoCo XML model: the body
tree (``pos``/``quat``), hinge ``axis``/``range``/``limited``, the free root
joint (or the planar base of galaxea_r1pro: slide x, slide y, hinge z on the root body), ``<include>``,
``<compiler angle=... autolimits=...>`` and default-class inheritance for joint attributes.  Inertials, geoms, meshes, actuators,
sensors and keyframes are ignored.

It replaces two reference loaders at once:

* ``mj.MjModel.from_xml_path`` as used by the IK side
  (reference general_motion_retargeting/motion_retarget.py:27) -- MuJoCo body
  order is depth first, ``nq = 7 + n_hinge``, ``nv = 6 + n_hinge``, body quats
  and hinge axes are normalised at compile time, ``qpos0`` is the root body
  ``pos`` + identity quaternion + zeros.
* ``KinematicsModel._parse_xml`` (reference kinematics_model.py:101-163) --
  same depth-first order, xyzw local rotations kept *raw* (not normalised),
  limits as float32.

Both views are kept on one :class:`RobotModel`.
"""
from __future__ import annotations

import dataclasses
import json
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

JNT_NONE = 0
JNT_HINGE = 1
JNT_FREE = 2

ROOT_DOFS_FREE = 0x3F    # x y z rx ry rz
ROOT_DOFS_PLANAR = 0x23  # x y . . . rz -- a mobile base written as slide x + slide y + hinge z on the root body


class MjcfError(ValueError):
    pass


@dataclasses.dataclass
class BodyView:
    """What ``mujoco.MjModel.body(name)`` is used for in GMR's callers: ``.id`` and ``.name``."""
    id: int
    name: str


@dataclasses.dataclass
class RobotModel:
    """Kinematic tree of one robot (all arrays are numpy, float64 unless noted).

    ``body_quat_raw`` is the XML value (wxyz) untouched -- the reference torch FK
    uses it as is (kinematics_model.py:119-123).  ``body_quat`` is the unit
    quaternion MuJoCo would use.
    """

    name: str
    source: str
    body_names: List[str]
    parent: np.ndarray          # int32 [nb], -1 for the root
    body_pos: np.ndarray        # [nb,3]
    body_quat: np.ndarray       # [nb,4] wxyz, unit
    body_quat_raw: np.ndarray   # [nb,4] wxyz, as written
    jnt_type: np.ndarray        # int32 [nb]
    jnt_axis: np.ndarray        # [nb,3] unit (0 for bodies without a hinge)
    jnt_axis_raw: np.ndarray    # [nb,3] as written
    jnt_range: np.ndarray       # [nb,2]
    jnt_limited: np.ndarray     # bool [nb]
    jnt_names: List[Optional[str]]
    qpos_adr: np.ndarray        # int32 [nb], -1 if no joint
    dof_adr: np.ndarray         # int32 [nb], -1 if no joint
    nq: int
    nv: int
    timestep: float
    angle_unit: str
    # Which of the root's six free-joint dofs exist.  Internally every robot carries a free-joint root (``nq = 7 + hinges``): a
    # planar base is that joint with z / roll / pitch taken out of the IK (ROOT_DOFS_PLANAR), and ``to_mj_qpos`` / ``from_mj_qpos``
    # translate to MuJoCo's ``[x, y, yaw, hinges]``.
    root_dofs: int = ROOT_DOFS_FREE
    root_jnt_names: Optional[List[str]] = None

    # ------------------------------------------------------------------
    @property
    def planar_base(self) -> bool:
        return self.root_dofs == ROOT_DOFS_PLANAR

    @property
    def mj_nq(self) -> int:
        """``mujoco.MjModel.nq`` of the XML: 3 instead of 7 coordinates for a planar base."""
        return self.nq - 4 if self.planar_base else self.nq

    @property
    def mj_nv(self) -> int:
        return self.nv - 3 if self.planar_base else self.nv

    def to_mj_qpos(self, q, yaw_ref=None):
        """Internal ``[..., nq]`` -> the XML's qpos layout (numpy or torch; identity for a free root).  A planar base gives
        ``[x - x0, y - y0, yaw, hinges]``; ``yaw_ref`` (the previous frame's yaw, same leading shape) picks the branch of the
        angle nearest to it -- MuJoCo's hinge coordinate accumulates, a quaternion does not."""
        if not self.planar_base:
            return q
        import math
        xp = __import__("torch") if type(q).__module__.startswith("torch") else np
        yaw = 2.0 * xp.arctan2(q[..., 6], q[..., 3])
        yaw = xp.where(yaw > math.pi, yaw - 2 * math.pi, xp.where(yaw <= -math.pi, yaw + 2 * math.pi, yaw))
        if yaw_ref is not None:
            yaw = yaw + 2 * math.pi * xp.round((yaw_ref - yaw) / (2 * math.pi))
        head = xp.stack([q[..., 0] - float(self.body_pos[0, 0]), q[..., 1] - float(self.body_pos[0, 1]), yaw], -1)
        return xp.concatenate([head, q[..., 7:]], -1) if xp is np else xp.cat([head, q[..., 7:]], -1)

    def from_mj_qpos(self, q):
        """The XML's qpos layout -> internal (numpy)."""
        if not self.planar_base:
            return np.asarray(q, dtype=np.float64)
        q = np.asarray(q, dtype=np.float64)
        out = np.zeros(q.shape[:-1] + (self.nq,))
        out[..., 0] = q[..., 0] + self.body_pos[0, 0]
        out[..., 1] = q[..., 1] + self.body_pos[0, 1]
        out[..., 2] = self.body_pos[0, 2]
        out[..., 3] = np.cos(0.5 * q[..., 2])
        out[..., 6] = np.sin(0.5 * q[..., 2])
        out[..., 7:] = q[..., 3:]
        return out

    @property
    def nbody(self) -> int:
        return len(self.body_names)

    @property
    def ndof_hinge(self) -> int:
        return int((self.jnt_type == JNT_HINGE).sum())

    @property
    def qpos0(self) -> np.ndarray:
        q = np.zeros(self.nq)
        q[0:3] = self.body_pos[0]
        q[3:7] = self.body_quat[0]
        return q

    def body_index(self, name: str) -> int:
        try:
            return self.body_names.index(name)
        except ValueError:
            raise KeyError(name) from None

    # --- the two ``mujoco.MjModel`` accessors GMR's callers use (scripts/fbx_to_robot.py:1040-1041, 1157-1158): MuJoCo ids
    #     count the world body as 0, so a robot body's id is its index here + 1 -- the row of ``configuration.data.xpos``.
    def body(self, key) -> "BodyView":
        if isinstance(key, str):
            if key == "world":
                return BodyView(0, "world")
            return BodyView(self.body_index(key) + 1, key)
        i = int(key)
        if not 0 <= i <= self.nbody:
            raise IndexError(i)
        return BodyView(i, "world" if i == 0 else self.body_names[i - 1])

    def name2id(self, name: str) -> int:
        """``mujoco.mj_name2id(model, mjOBJ_BODY, name)``: -1 for an unknown name."""
        try:
            return self.body(name).id
        except KeyError:
            return -1

    @property
    def depth(self) -> np.ndarray:
        d = np.zeros(self.nbody, dtype=np.int32)
        for b in range(1, self.nbody):
            d[b] = d[self.parent[b]] + 1
        return d

    def hinge_bodies(self) -> np.ndarray:
        return np.nonzero(self.jnt_type == JNT_HINGE)[0].astype(np.int32)

    def dof_limits(self):
        """(lower, upper) per hinge in dof order, radians."""
        hb = self.hinge_bodies()
        return self.jnt_range[hb, 0].copy(), self.jnt_range[hb, 1].copy()

    # ------------------------------------------------------------------
    def to_dict(self) -> dict:
        return {
            "format": "gmr_amd.robot.v1",
            "name": self.name,
            "source": self.source,
            "timestep": self.timestep,
            "angle_unit": self.angle_unit,
            "nq": self.nq,
            "nv": self.nv,
            "bodies": [
                {
                    "name": self.body_names[b],
                    "parent": int(self.parent[b]),
                    "pos": [float(x) for x in self.body_pos[b]],
                    "quat": [float(x) for x in self.body_quat_raw[b]],
                    "joint": None
                    if self.jnt_type[b] == JNT_NONE
                    else {
                        "type": ("planar" if self.planar_base else "free") if self.jnt_type[b] == JNT_FREE else "hinge",
                        "name": self.jnt_names[b],
                        **({"names": list(self.root_jnt_names or [])} if self.jnt_type[b] == JNT_FREE and self.planar_base else {}),
                        "axis": [float(x) for x in self.jnt_axis_raw[b]],
                        "range": [float(x) for x in self.jnt_range[b]],
                        "limited": bool(self.jnt_limited[b]),
                    },
                }
                for b in range(self.nbody)
            ],
        }

    @staticmethod
    def from_dict(d: dict) -> "RobotModel":
        if d.get("format") != "gmr_amd.robot.v1":
            raise MjcfError("not a gmr_amd robot pack")
        recs = []
        for b in d["bodies"]:
            j = b["joint"]
            recs.append(
                _BodyRec(
                    name=b["name"],
                    parent=b["parent"],
                    pos=np.asarray(b["pos"], dtype=np.float64),
                    quat=np.asarray(b["quat"], dtype=np.float64),
                    jtype=JNT_NONE if j is None else (JNT_FREE if j["type"] in ("free", "planar") else JNT_HINGE),
                    root_dofs=ROOT_DOFS_PLANAR if j is not None and j["type"] == "planar" else ROOT_DOFS_FREE,
                    root_names=None if j is None or j["type"] != "planar" else list(j.get("names", [])),
                    jname=None if j is None else j["name"],
                    axis=np.zeros(3) if j is None else np.asarray(j["axis"], dtype=np.float64),
                    rng=np.zeros(2) if j is None else np.asarray(j["range"], dtype=np.float64),
                    limited=False if j is None else bool(j["limited"]),
                )
            )
        return _assemble(d["name"], d["source"], recs, d["timestep"], d["angle_unit"])


@dataclasses.dataclass
class _BodyRec:
    name: str
    parent: int
    pos: np.ndarray
    quat: np.ndarray
    jtype: int
    jname: Optional[str]
    axis: np.ndarray
    rng: np.ndarray
    limited: bool
    root_dofs: int = ROOT_DOFS_FREE
    root_names: Optional[List[str]] = None


def _floats(s: str, n: int, what: str) -> np.ndarray:
    v = np.array([float(t) for t in s.split()], dtype=np.float64)
    if v.shape[0] != n:
        raise MjcfError(f"{what}: expected {n} numbers, got {s!r}")
    return v


def _expand_includes(elem: ET.Element, base_dir: str, root_dir: str, depth: int = 0) -> None:
    """Splice ``<include file=...>`` children in place (recursively)."""
    if depth > 16:
        raise MjcfError("include nesting too deep")
    i = 0
    while i < len(elem):
        child = elem[i]
        if child.tag == "include":
            fn = child.attrib.get("file")
            if fn is None:
                raise MjcfError("<include> without file")
            path = os.path.join(base_dir, fn)
            if not os.path.exists(path):
                path = os.path.join(root_dir, fn)
            inc_root = ET.parse(path).getroot()
            _expand_includes(inc_root, os.path.dirname(path), root_dir, depth + 1)
            elem.remove(child)
            for k, sub in enumerate(list(inc_root)):
                elem.insert(i + k, sub)
            i += len(inc_root)
        else:
            _expand_includes(child, base_dir, root_dir, depth)
            i += 1


class _Defaults:
    """Joint attribute defaults by class (nested ``<default class=...>``)."""

    def __init__(self, root: ET.Element):
        self.by_class: Dict[str, Dict[str, str]] = {"main": {}}
        for top in root.findall("default"):
            self._walk(top, top.attrib.get("class", "main"), {})

    def _walk(self, node: ET.Element, cls: str, inherited: Dict[str, str]) -> None:
        attrs = dict(inherited)
        for j in node.findall("joint"):
            attrs.update(j.attrib)
        merged = dict(self.by_class.get(cls, {}))
        merged.update(attrs)
        self.by_class[cls] = merged
        for sub in node.findall("default"):
            sub_cls = sub.attrib.get("class")
            if sub_cls is None:
                raise MjcfError("nested <default> without class")
            self._walk(sub, sub_cls, attrs)

    def joint_attrs(self, elem: ET.Element, childclass: Optional[str]) -> Dict[str, str]:
        cls = elem.attrib.get("class", childclass or "main")
        out = dict(self.by_class.get("main", {})) if cls not in self.by_class else dict(self.by_class[cls])
        out.update(elem.attrib)
        return out


def load_mjcf(path: str, name: Optional[str] = None) -> RobotModel:
    """Compile the kinematic subset of an MJCF file into a :class:`RobotModel`."""
    path = os.fspath(path)
    root = ET.parse(path).getroot()
    root_dir = os.path.dirname(os.path.abspath(path))
    _expand_includes(root, root_dir, root_dir)

    comp = {}
    for c in root.findall("compiler"):
        comp.update(c.attrib)
    angle_unit = comp.get("angle", "degree")
    if angle_unit not in ("degree", "radian"):
        raise MjcfError(f"bad compiler angle {angle_unit!r}")
    autolimits = comp.get("autolimits", "true") == "true"
    timestep = 0.002
    for o in root.findall("option"):
        if "timestep" in o.attrib:
            timestep = float(o.attrib["timestep"])

    defaults = _Defaults(root)

    robot_roots = [b for wb in root.findall("worldbody") for b in wb.findall("body")]
    if len(robot_roots) != 1:
        raise MjcfError(f"expected exactly one top-level <body>, found {len(robot_roots)}")

    recs: List[_BodyRec] = []

    def add_body(node: ET.Element, parent: int, childclass: Optional[str]) -> None:
        childclass = node.attrib.get("childclass", childclass)
        for k in ("euler", "axisangle", "xyaxes", "zaxis"):
            if k in node.attrib:
                raise MjcfError(f"body {node.attrib.get('name')}: orientation attribute {k!r} not supported")
        pos = _floats(node.attrib.get("pos", "0 0 0"), 3, "body pos")
        quat = _floats(node.attrib.get("quat", "1 0 0 0"), 4, "body quat")
        joints = [(j, True) for j in node.findall("freejoint")] + [(j, False) for j in node.findall("joint")]
        jtype, jname, axis, rng, limited = JNT_NONE, None, np.zeros(3), np.zeros(2), False
        root_dofs, root_names = ROOT_DOFS_FREE, None
        if parent == -1 and len(joints) == 3 and not any(f for _, f in joints):
            # a planar mobile base (assets/galaxea_r1pro/r1_pro.xml:102-104): slide x, slide y, hinge z, in this order, unlimited
            want = [("slide", (1.0, 0.0, 0.0)), ("slide", (0.0, 1.0, 0.0)), ("hinge", (0.0, 0.0, 1.0))]
            root_names = []
            for (jel, _), (typ, ax) in zip(joints, want):
                attrs = defaults.joint_attrs(jel, childclass)
                a = _floats(attrs.get("axis", "0 0 1"), 3, "joint axis")
                lim = attrs.get("limited", "auto")
                if attrs.get("type", "hinge") != typ or tuple(a) != ax or "range" in attrs or lim == "true" or \
                        np.any(_floats(attrs.get("pos", "0 0 0"), 3, "joint pos") != 0.0) or float(attrs.get("ref", 0.0)) != 0.0:
                    raise MjcfError("root body with three joints: only slide x, slide y, hinge z (unlimited, at the body origin) is supported")
                root_names.append(attrs.get("name"))
            if np.any(quat != np.array([1.0, 0.0, 0.0, 0.0])):
                raise MjcfError("planar base: the root body must not be rotated")
            jtype, jname, root_dofs = JNT_FREE, root_names[0], ROOT_DOFS_PLANAR
            joints = []
        elif len(joints) > 1:
            raise MjcfError(f"body {node.attrib.get('name')}: {len(joints)} joints on one body not supported")
        if joints:
            jel, is_freejoint = joints[0]
            attrs = dict(jel.attrib) if is_freejoint else defaults.joint_attrs(jel, childclass)
            typ = "free" if is_freejoint else attrs.get("type", "hinge")
            jname = attrs.get("name")
            if typ == "free":
                if parent != -1:
                    raise MjcfError("free joint below the root body")
                jtype = JNT_FREE
            elif typ == "hinge":
                if parent == -1:
                    raise MjcfError("robot root must carry a free joint (fixed-base/hinge root not supported)")
                jtype = JNT_HINGE
                jpos = _floats(attrs.get("pos", "0 0 0"), 3, "joint pos")
                if np.any(jpos != 0.0):
                    raise MjcfError(f"joint {jname}: non-zero joint pos not supported")
                if "ref" in attrs and float(attrs["ref"]) != 0.0:
                    raise MjcfError(f"joint {jname}: ref not supported")
                axis = _floats(attrs.get("axis", "0 0 1"), 3, "joint axis")
                if "range" in attrs:
                    rng = _floats(attrs["range"], 2, "joint range")
                    if angle_unit == "degree":
                        rng = np.deg2rad(rng)
                lim = attrs.get("limited", "auto")
                if lim == "true":
                    limited = True
                elif lim == "false":
                    limited = False
                else:
                    limited = bool(autolimits and "range" in attrs and rng[0] < rng[1])
            else:
                raise MjcfError(f"joint {jname}: type {typ!r} not supported (hinge/free only)")
        elif parent == -1 and jtype != JNT_FREE:
            raise MjcfError("robot root must carry a free joint")
        idx = len(recs)
        recs.append(_BodyRec(node.attrib.get("name", f"body{idx}"), parent, pos, quat, jtype, jname, axis, rng, limited, root_dofs, root_names))
        for child in node.findall("body"):
            add_body(child, idx, childclass)

    add_body(robot_roots[0], -1, None)
    return _assemble(name or root.attrib.get("model", os.path.basename(path)), os.path.basename(path), recs, timestep, angle_unit)


def _assemble(name: str, source: str, recs: List[_BodyRec], timestep: float, angle_unit: str) -> RobotModel:
    nb = len(recs)
    parent = np.array([r.parent for r in recs], dtype=np.int32)
    pos = np.stack([r.pos for r in recs])
    quat_raw = np.stack([r.quat for r in recs])
    qn = np.linalg.norm(quat_raw, axis=1, keepdims=True)
    if np.any(qn < 1e-12):
        raise MjcfError("zero body quaternion")
    quat = quat_raw / qn
    jtype = np.array([r.jtype for r in recs], dtype=np.int32)
    axis_raw = np.stack([r.axis for r in recs])
    axis = axis_raw.copy()
    for b in range(nb):
        if jtype[b] == JNT_HINGE:
            n = np.linalg.norm(axis[b])
            if n < 1e-12:
                raise MjcfError(f"zero hinge axis on {recs[b].name}")
            axis[b] /= n
    rng = np.stack([r.rng for r in recs])
    limited = np.array([r.limited for r in recs], dtype=bool)
    qadr = np.full(nb, -1, dtype=np.int32)
    dadr = np.full(nb, -1, dtype=np.int32)
    nq = nv = 0
    for b in range(nb):
        if jtype[b] == JNT_FREE:
            qadr[b], dadr[b] = nq, nv
            nq, nv = nq + 7, nv + 6
        elif jtype[b] == JNT_HINGE:
            qadr[b], dadr[b] = nq, nv
            nq, nv = nq + 1, nv + 1
    if jtype[0] != JNT_FREE:
        raise MjcfError("root body has no free joint")
    return RobotModel(
        name=name, source=source, body_names=[r.name for r in recs], parent=parent, body_pos=pos,
        body_quat=quat, body_quat_raw=quat_raw, jnt_type=jtype, jnt_axis=axis, jnt_axis_raw=axis_raw,
        jnt_range=rng, jnt_limited=limited, jnt_names=[r.jname for r in recs], qpos_adr=qadr, dof_adr=dadr,
        nq=nq, nv=nv, timestep=timestep, angle_unit=angle_unit, root_dofs=recs[0].root_dofs, root_jnt_names=recs[0].root_names,
    )


def load_robot(path: str, name: Optional[str] = None) -> RobotModel:
    """Load either an MJCF ``.xml`` or a ``gmr_amd.robot.v1`` JSON pack."""
    path = os.fspath(path)
    if path.endswith(".xml"):
        return load_mjcf(path, name)
    with open(path) as f:
        d = json.load(f)
    return RobotModel.from_dict(d)
