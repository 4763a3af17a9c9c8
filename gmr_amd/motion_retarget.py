"""``GeneralMotionRetargeting`` -- GMR's retarget API on the MI355X engine.

Same constructor, ``retarget()`` signature, statefulness and error behaviour as the
reference class (general_motion_retargeting/motion_retarget.py:10-270), so it drops in under
``scripts/*_to_robot_dataset.py``; plus ``retarget_batch()`` which hands whole clips (or many
clips) to the frames-batched kernel in one launch.

What is different by construction: the per-frame ``mink.solve_ik`` / MuJoCo / DAQP stack is
replaced by libgmr_amd.so's HIP kernels (exact box-QP, so ``solver`` only names what the
reference would have used); ``self.model`` is this package's ``RobotModel`` (kinematic subset
of the MJCF) rather than a ``mujoco.MjModel``; ``self.configuration`` is a small view object
exposing ``.q`` / ``.data.qpos``.  There is no CPU fallback: without the native library and
a HIP device the constructor raises.
"""
from __future__ import annotations

from typing import List, Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from ._native import IKParams
from .engine import Engine
from .ik_config import load_ik_config
from .mjcf import load_robot
from .model import compile_model
from .params import IK_CONFIG_DICT, ROBOT_XML_DICT
from .schedule import make_items


class _Data:
    def __init__(self, owner):
        self._o = owner

    @property
    def qpos(self) -> np.ndarray:
        return self._o._caller_qpos().copy()

    @property
    def xpos(self) -> np.ndarray:
        """MuJoCo ``data.xpos`` at the current configuration: ``[nbody + 1, 3]``, row 0 the world body (zeros), row
        ``model.body(name).id`` the named body -- the indexing scripts/fbx_to_robot.py:1040-1041, 1157-1158 uses."""
        x = self._o._engine.evaluate(self._o._state, want_errors=False, want_poses=True)[1].cpu().numpy()[0]
        return np.concatenate([np.zeros((1, 3)), x])

    @property
    def xquat(self) -> np.ndarray:
        x = self._o._engine.evaluate(self._o._state, want_errors=False, want_poses=True)[2].cpu().numpy()[0]
        return np.concatenate([np.array([[1.0, 0.0, 0.0, 0.0]]), x])


class _Configuration:
    """Stand-in for ``mink.Configuration``: current generalized coordinates only."""

    def __init__(self, owner):
        self._o = owner
        self.model = owner.model
        self.data = _Data(owner)

    @property
    def q(self) -> np.ndarray:
        return self.data.qpos.copy()


class _FrameTask:
    """Stand-in for one ``mink.FrameTask`` of ``tasks1`` / ``tasks2`` (motion_retarget.py:83-89, 101-107): the attributes
    callers read -- ``frame_name``, ``frame_type``, the two costs -- and ``compute_error(configuration)``, served by
    ``gmr_evaluate`` at the object's current configuration and last targets (scripts/fbx_to_robot.py:1129-1132)."""

    frame_type = "body"
    lm_damping = 1.0
    gain = 1.0

    def __init__(self, owner, table: int, index: int, task):
        self._o, self._table, self._index = owner, table, index
        self.frame_name = task.frame
        self.human_body = task.human
        self.position_cost, self.orientation_cost = task.pos_weight, task.rot_weight

    def compute_error(self, configuration=None) -> np.ndarray:
        """``Log(T_body^-1 T_target)`` as ``[v; w]`` (the sign convention differs between mink releases; norms do not)."""
        o = self._o
        if configuration is not None and configuration is not o.configuration:
            raise ValueError("tasks are bound to their retargeter's configuration")
        return o._task_errors()[(o._cm_ntask0 if self._table else 0) + self._index].copy()

    def __repr__(self):
        return f"FrameTask(frame_name={self.frame_name!r}, position_cost={self.position_cost}, orientation_cost={self.orientation_cost})"

    def __eq__(self, other):  # round 1 exposed plain frame-name strings: keep `t == "pelvis"` and `"pelvis" in tasks1` working
        if isinstance(other, str):
            return self.frame_name == other
        return self is other

    def __hash__(self):
        return hash(self.frame_name)


class GeneralMotionRetargeting:
    """General Motion Retargeting (GMR) on gfx950."""

    HOST_PIPELINE_MIN_FRAMES = 1 << 18  # numpy batches at least this long go through the overlapped host pipeline

    def __init__(
        self,
        src_human: str,
        tgt_robot: str,
        actual_human_height: float = None,
        solver: str = "daqp",
        damping: float = 5e-1,
        verbose: bool = False,
        device: int = 0,
        persistent_session_ms: int = 0,
    ) -> None:
        # persistent_session_ms > 0 (not in the reference's signature): retarget() frames go through a resident wavefront that idles
        # out after that many milliseconds (gmr_session_set_persistent) -- for a process whose job is the live loop
        self._persistent_ms = int(persistent_session_ms)
        # robot model (motion_retarget.py:24-27)
        self.xml_file = str(ROBOT_XML_DICT[tgt_robot])
        if verbose:
            print("Use robot model: ", self.xml_file)
        self.model = load_robot(self.xml_file, name=tgt_robot)
        # IK config (:30-33)
        cfg_path = IK_CONFIG_DICT[src_human][tgt_robot]
        ik_config = load_ik_config(cfg_path)
        if verbose:
            print("Use IK config: ", cfg_path)
        self._cm = compile_model(self.model, ik_config, actual_human_height)
        ratio = self._cm.ratio

        self.ik_match_table1 = ik_config.ik_match_table1
        self.ik_match_table2 = ik_config.ik_match_table2
        self.human_root_name = ik_config.human_root_name
        self.robot_root_name = ik_config.robot_root_name
        self.use_ik_match_table1 = ik_config.use_ik_match_table1
        self.use_ik_match_table2 = ik_config.use_ik_match_table2
        self.human_scale_table = {k: v * ratio for k, v in ik_config.human_scale_table.items()}  # :42-43
        self.ground = ik_config.ground_height * np.array([0, 0, 1])
        self.max_iter = 10
        self.solver = solver
        self.damping = damping

        # offsets keyed by human body name, non-zero-weight entries only (:80-114)
        self.pos_offsets1, self.rot_offsets1, self.pos_offsets2, self.rot_offsets2 = {}, {}, {}, {}
        for tab, po, ro in ((ik_config.table1, self.pos_offsets1, self.rot_offsets1), (ik_config.table2, self.pos_offsets2, self.rot_offsets2)):
            for t in tab:
                if t.pos_weight != 0 or t.rot_weight != 0:
                    po[t.human] = np.array(t.pos_offset) - self.ground
                    ro[t.human] = R.from_quat(t.rot_offset, scalar_first=True)
        self._cm_ntask0 = len(self._cm.tasks[0])
        self.tasks1 = [_FrameTask(self, 0, i, t) for i, t in enumerate(self._cm.tasks[0])]
        self.tasks2 = [_FrameTask(self, 1, i, t) for i, t in enumerate(self._cm.tasks[1])]
        self.human_body_to_task1 = {t.human_body: t for t in self.tasks1}  # keyed by human body name (:90,108)
        self.human_body_to_task2 = {t.human_body: t for t in self.tasks2}

        self._engine = Engine(self._cm, device)
        self.device = self._engine.device
        self.setup_retarget_configuration()

    # ------------------------------------------------------------------ state
    def setup_retarget_configuration(self):
        """Reset to ``qpos0`` (what a fresh ``mink.Configuration(model)`` holds, :75)."""
        self._qpos = np.array(self.model.qpos0, dtype=np.float64)  # the current configuration (host copy, free-joint layout)
        self._yaw = 0.0  # planar base only: the accumulated heading MuJoCo's hinge coordinate would hold (RobotModel.to_mj_qpos)
        self.configuration = _Configuration(self)
        self.scaled_human_data = None
        self._last_pos_np = None
        self._col_cache: Dict[Tuple[str, ...], np.ndarray] = {}
        for s in getattr(self, "_sessions", {}).values():
            s.close()
        self._sessions: Dict[tuple, object] = {}  # live single-sequence sessions, one per (column layout, solver settings)
        self._session_key = None

    @property
    def _state(self) -> torch.Tensor:
        return torch.from_numpy(self._qpos).to(self.device).reshape(1, -1)

    def _caller_qpos(self) -> np.ndarray:
        """The current configuration in the XML's own qpos layout: ``[x, y, z, qw..qz, hinges]`` for a free-joint root (the
        engine's layout), ``[x, y, yaw, hinges]`` for the planar base of galaxea_r1pro."""
        if not self.model.planar_base:
            return self._qpos
        return self.model.to_mj_qpos(self._qpos, yaw_ref=np.float64(self._yaw))

    def _caller_layout_batch(self, out: torch.Tensor, offs: np.ndarray) -> torch.Tensor:
        """``[N, nq]`` engine output -> the XML's layout; a planar base's heading is unwrapped along every clip from its first
        frame's principal value (the hinge coordinate of the reference accumulates from ``qpos0``'s 0)."""
        if not self.model.planar_base or out.shape[0] == 0:
            return out
        mj = self.model.to_mj_qpos(out)
        yaw = mj[:, 2]
        jump = torch.round((yaw[1:] - yaw[:-1]) / (2 * np.pi))
        starts = torch.as_tensor(np.asarray(offs[:-1], dtype=np.int64), device=out.device)
        starts = starts[starts < out.shape[0]]
        inner = starts[starts > 0]
        jump[inner - 1] = 0  # no unwrapping across a clip boundary
        c = torch.cat([torch.zeros(1, dtype=yaw.dtype, device=yaw.device), torch.cumsum(jump, 0)])
        lens = torch.diff(torch.cat([starts, torch.tensor([out.shape[0]], device=out.device)]))
        base = torch.repeat_interleave(c[starts], lens)
        mj[:, 2] = yaw - 2 * np.pi * (c - base)
        return mj

    def _session(self, names: Sequence[str]):
        """The live session for this frame layout; the warm start follows the object, not the session."""
        key = (tuple(names), float(self.damping), int(self.max_iter))
        s = self._sessions.get(key)
        if s is None:
            s = self._engine.session(self._columns(names), len(names), self._params(False))
            if self._persistent_ms > 0:
                s.set_persistent(self._persistent_ms)
            self._sessions[key] = s
        if key != self._session_key:
            s.reset(self._qpos)
            self._session_key = key
        return s

    def _params(self, offset_to_ground: bool) -> IKParams:
        return IKParams(damping=self.damping, max_iter=self.max_iter, offset_to_ground=int(bool(offset_to_ground)))

    @property
    def ik_columns(self) -> List[str]:
        """The human bodies this (source, robot) config consumes (``human_scale_table`` keys that carry a table-1 task,
        motion_retarget.py:209-250), in the order the solver holds them: what ``columns=`` of the input adapters takes, so that
        ``retarget_batch`` reads a dense ``[N, len(ik_columns), 7]`` instead of picking these columns out of 24 or 55."""
        return list(self._cm.slot_names)

    def _columns(self, names: Sequence[str]) -> np.ndarray:
        key = tuple(names)
        if key not in self._col_cache:
            self._col_cache[key] = self._cm.slot_columns(names)
        return self._col_cache[key]

    # ------------------------------------------------------------------ per-frame API (drop-in)
    def update_targets(self, human_data, offset_to_ground=False):
        """Host-side target preparation, kept for callers that read ``scaled_human_data`` (:117-124).
        The solver itself prepares targets on the GPU from the raw key-points."""
        human_data = self.to_numpy(human_data)
        names = list(human_data.keys())
        self._last_cols = self._columns(names)
        self._last_pos_np = np.stack([np.asarray(human_data[n][0], dtype=np.float64).reshape(3) for n in names])
        self._last_quat_np = np.stack([np.asarray(human_data[n][1], dtype=np.float64).reshape(4) for n in names])
        self._last_offset_to_ground = offset_to_ground
        human_data = self.scale_human_data(human_data, self.human_root_name, self.human_scale_table)
        human_data = self.offset_human_data(human_data, self.pos_offsets1, self.rot_offsets1)
        if offset_to_ground:
            human_data = self.offset_human_data_to_ground(human_data)
        self.scaled_human_data = human_data

    def retarget(self, human_data, offset_to_ground=False):
        """One frame, warm-started from the previous call; returns a fresh ``qpos`` copy (:139-185)."""
        human_data = self.to_numpy(human_data)  # mutates the caller's dict like the reference (:203-206)
        names = list(human_data.keys())
        cols = self._columns(names)  # KeyError exactly where the reference raises
        pos = np.stack([np.asarray(human_data[n][0], dtype=np.float64).reshape(3) for n in names])
        quat = np.stack([np.asarray(human_data[n][1], dtype=np.float64).reshape(4) for n in names])
        self._last_human_data = human_data
        self._last_offset_to_ground = offset_to_ground
        self.scaled_human_data = _LazyScaled(self)
        self._last_pos_np, self._last_quat_np, self._last_cols = pos, quat, cols
        q, solves = self._session(names).step(pos, quat, offset_to_ground)
        self.last_num_solves = solves & 0x3FFFFFFF
        if not np.all(np.isfinite(q)):
            raise FloatingPointError("retarget produced non-finite qpos")
        self._qpos = q
        if self.model.planar_base:
            self._yaw = float(self.model.to_mj_qpos(q, yaw_ref=np.float64(self._yaw))[2])
        return self._caller_qpos().copy()

    def _evaluate(self, want_task_errors: bool):
        if getattr(self, "_last_pos_np", None) is None:
            raise RuntimeError("no targets set: call retarget() or update_targets() first")  # mink raises TargetNotSet
        pos = torch.from_numpy(self._last_pos_np[None]).to(self.device)
        quat = torch.from_numpy(self._last_quat_np[None]).to(self.device)
        return self._engine.evaluate(self._state, pos, quat, self._last_cols, offset_to_ground=self._last_offset_to_ground,
                                     want_task_errors=want_task_errors)

    def _errors(self) -> np.ndarray:
        return self._evaluate(False)[0].cpu().numpy()[0]

    def _task_errors(self) -> np.ndarray:
        """[ntask1 + ntask2, 6]: ``task.compute_error(configuration)`` of every task, table 1 first."""
        return self._evaluate(True)[3].cpu().numpy()[0]

    def error1(self):
        """|concat of the table-1 task errors| at the current configuration and targets (:188-193)."""
        return float(self._errors()[0])

    def error2(self):
        return float(self._errors()[1])

    # ------------------------------------------------------------------ batched API
    def retarget_batch(self, pos, quat, body_names: Sequence[str], seq_offsets=None, chunk=0, burn_in: int = 0,
                       offset_to_ground: bool = False, return_iters: bool = False, verify: bool = True,
                       human_heights: Optional[Sequence[float]] = None, check: bool = True, clip_start: str = "qpos0"):
        """Retarget whole clips in one launch.

        pos ``[N, B, 3]`` (m), quat ``[N, B, 4]`` (wxyz), float32/float64, numpy or CUDA torch; ``body_names`` names the
        B columns; ``seq_offsets [S+1]`` delimits independent clips (default: one clip).  Every clip starts from
        ``qpos0`` like a fresh reference object.  ``chunk > 0`` (or ``chunk="auto"``: chunk and burn-in chosen from the clip lengths,
        ``schedule.auto_chunk``) solves each clip in parallel-in-time chunks; with ``verify``
        (default) chunk boundaries are checked and repaired so the result equals the sequential run to 1e-7
        (``Engine.ik_solve_chunked``); ``verify=False`` is the raw burn-in approximation (schedule.py).
        ``human_heights [S]`` gives every clip its own ``actual_human_height`` -- what the reference does by building one
        retargeter per file (scripts/smplx_to_robot_dataset.py:79-83): clip s is solved with the scale table of
        ``GeneralMotionRetargeting(src, robot, human_heights[s])`` whatever height this object was built with.  With ``check``
        (default) the result is inspected on the device -- non-finite qpos raises ``FloatingPointError`` (as ``retarget`` does),
        a QP that hit its iteration cap raises ``RuntimeError`` (mink asserts on a failed QP); this synchronises the stream.
        ``clip_start="root_target"`` is an opt-in departure from the reference: every clip starts with the floating base on its
        first frame's root target instead of ``qpos0`` at the world origin, which avoids the reference's slow -- for clips facing
        away from ``qpos0`` sometimes never-ending -- start-up; the default reproduces the reference.
        Returns qpos ``[N, nq]`` float64 (same container kind as the input) and, optionally, solves per frame.

        Memory: numpy batches of at least ``HOST_PIPELINE_MIN_FRAMES`` (2^18) frames go through the overlapped host pipeline
        (``Engine.ik_solve_host``) and come back as a numpy view of PAGE-LOCKED memory the kernel wrote directly; it stays pinned as
        long as the array (or any slice of it) is alive -- copy what must outlive the batch.  Planar-base robots (galaxea_r1pro,
        qpos ``[x, y, yaw, hinges]``): the yaw is unwrapped along each clip starting from the first frame's principal value in
        (-pi, pi]; MuJoCo's hinge coordinate accumulates from 0 instead, so for a clip that starts facing backwards the two can
        differ by a multiple of 2 pi (same pose; no reference fixture pins either branch).
        """
        is_np = isinstance(pos, np.ndarray)
        if clip_start not in ("qpos0", "root_target"):
            raise ValueError("clip_start must be 'qpos0' (the reference) or 'root_target'")
        from ._native import INIT_QPOS0, INIT_ROOT_TARGET
        clip_init = INIT_ROOT_TARGET if clip_start == "root_target" else INIT_QPOS0
        cols = self._columns(list(body_names))  # KeyError where the reference raises
        if is_np and isinstance(quat, np.ndarray) and not isinstance(chunk, str) and chunk == 0 and clip_init == INIT_QPOS0 and pos.ndim == 3 and pos.shape[0] >= self.HOST_PIPELINE_MIN_FRAMES \
                and not self.model.planar_base:
            # big host batches: two streams, copies overlapped with the kernel, pinned result (Engine.ik_solve_host)
            N = int(pos.shape[0])
            offs = np.asarray([0, N] if seq_offsets is None else seq_offsets, dtype=np.int64)
            hs = None
            if human_heights is not None:
                hh = np.asarray(human_heights, dtype=np.float64)
                if hh.shape != (len(offs) - 1,):
                    raise ValueError("human_heights must hold one height per clip")
                hs = hh / self._cm.config.human_height_assumption / self._cm.ratio
            out, iters = self._engine.ik_solve_host(pos, quat, cols, offs, params=self._params(offset_to_ground), height_scales=hs,
                                                    want_iters=return_iters, check=check)
            return (out, iters) if return_iters else out
        if is_np and isinstance(quat, np.ndarray) and pos.ndim == 3 and pos.shape[1] > len(cols):
            # host arrays with more bodies than the config consumes (55 SMPL-X joints, 14 used): gather the used columns on the
            # host first -- a quarter of the bytes cross PCIe
            pos, quat = pos[:, cols], quat[:, cols]
            cols = np.arange(len(cols), dtype=np.int32)
        tpos = torch.from_numpy(np.ascontiguousarray(pos)) if is_np else pos
        tquat = torch.from_numpy(np.ascontiguousarray(quat)) if isinstance(quat, np.ndarray) else quat
        tpos, tquat = tpos.to(self.device), tquat.to(self.device)
        N = int(tpos.shape[0])
        if seq_offsets is None:
            seq_offsets = [0, N]
        offs = np.asarray(seq_offsets, dtype=np.int64)
        if offs[0] != 0 or offs[-1] != N:
            raise ValueError("seq_offsets must span [0, N]")
        hs = None
        if human_heights is not None:
            hh = np.asarray(human_heights, dtype=np.float64)
            if hh.shape != (len(offs) - 1,):
                raise ValueError("human_heights must hold one height per clip")
            # ratio of the clip / ratio compiled into the model (:36-43): the per-item factor on the scale table
            hs = hh / self._cm.config.human_height_assumption / self._cm.ratio
        if isinstance(chunk, str):
            if chunk != "auto":
                raise ValueError("chunk must be an integer or 'auto'")
            from .schedule import auto_chunk
            chunk, burn_in = auto_chunk(offs, 8 * torch.cuda.get_device_properties(self.device).multi_processor_count)
        if chunk > 0 and verify:
            out, iters, self.last_chunk_info = self._engine.ik_solve_chunked(
                tpos, tquat, cols, offs, chunk, burn_in, params=self._params(offset_to_ground), height_scales=hs, clip_init=clip_init)
        else:
            items = make_items(offs, chunk=chunk, burn_in=burn_in, height_scales=hs, clip_init=clip_init)
            out, iters, _ = self._engine.ik_solve(tpos, tquat, cols, items, params=self._params(offset_to_ground))
            self.last_chunk_info = {"chunks": len(items), "passes": 0, "resolved_frames": 0}
        if check and N > 0:
            bad = torch.stack([(iters >> 31).ne(0).any(), ((iters >> 30) & 1).ne(0).any()]).cpu().numpy()  # flag bits of the solve counts
            if bad[0]:
                raise FloatingPointError("retarget_batch produced non-finite qpos")
            if bad[1]:
                raise RuntimeError("a box QP hit its iteration cap (the reference would assert on a failed QP)")
        out = self._caller_layout_batch(out, offs)
        if is_np:
            out = out.cpu().numpy()
            iters = iters.cpu().numpy() if iters is not None else None
        return (out, iters) if return_iters else out

    def fk_batch(self, root_pos, root_rot_xyzw, dof_pos, want_rot: bool = False):
        """Batched FK in the ``KinematicsModel.forward_kinematics`` convention (float32, xyzw; kinematics_model.py:213-246) on this
        object's robot: ``root_pos [T,3]``, ``root_rot_xyzw [T,4]``, ``dof_pos [T,ndof]`` -> ``body_pos [T,nbody,3]`` (and
        ``body_rot [T,nbody,4]`` with ``want_rot``).  numpy in -> numpy out, CUDA tensors in -> CUDA tensors out."""
        is_np = isinstance(root_pos, np.ndarray)
        conv = lambda a: (torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a).to(self.device, torch.float32)
        bp, br = self._engine.fk(conv(root_pos), conv(root_rot_xyzw), conv(dof_pos), want_rot=want_rot)
        if is_np:
            bp, br = bp.cpu().numpy(), (br.cpu().numpy() if br is not None else None)
        return (bp, br) if want_rot else bp

    # ------------------------------------------------------------------ helpers mirrored from the reference
    def to_numpy(self, human_data):
        for body_name in human_data.keys():
            human_data[body_name] = [np.asarray(human_data[body_name][0]), np.asarray(human_data[body_name][1])]
        return human_data

    def scale_human_data(self, human_data, human_root_name, human_scale_table):
        """Root scaled about the world origin, other bodies about the root; unlisted bodies dropped (:209-232)."""
        root_pos, root_quat = human_data[human_root_name]
        scaled_root_pos = human_scale_table[human_root_name] * root_pos
        out = {human_root_name: (scaled_root_pos, root_quat)}
        for body_name in human_data.keys():
            if body_name not in human_scale_table or body_name == human_root_name:
                continue
            out[body_name] = ((human_data[body_name][0] - root_pos) * human_scale_table[body_name] + scaled_root_pos, human_data[body_name][1])
        return out

    def offset_human_data(self, human_data, pos_offsets, rot_offsets):
        """Rotation offset first, then the position offset in the updated local frame (:234-250)."""
        out = {}
        for body_name in human_data.keys():
            pos, quat = human_data[body_name]
            rot = R.from_quat(quat, scalar_first=True) * rot_offsets[body_name]
            out[body_name] = [pos + rot.apply(pos_offsets[body_name]), rot.as_quat(scalar_first=True)]
        return out

    def offset_human_data_to_ground(self, human_data):
        """Shift everything so the lowest foot sits 0.1 m above z = 0 (:252-270)."""
        lowest = np.inf
        for body_name in human_data.keys():
            if "Foot" not in body_name and "foot" not in body_name:
                continue
            lowest = min(lowest, human_data[body_name][0][2])
        return {k: [v[0] - np.array([0, 0, lowest]) + np.array([0, 0, 0.1]), v[1]] for k, v in human_data.items()}


class _LazyScaled(dict):
    """``scaled_human_data`` computed on first access (only viewers read it; smplx_to_robot.py:133)."""

    def __init__(self, owner: GeneralMotionRetargeting):
        super().__init__()
        self._owner, self._done = owner, False

    def _fill(self):
        if not self._done:
            o = self._owner
            d = o.scale_human_data(o._last_human_data, o.human_root_name, o.human_scale_table)
            d = o.offset_human_data(d, o.pos_offsets1, o.rot_offsets1)
            if o._last_offset_to_ground:
                d = o.offset_human_data_to_ground(d)
            super().update(d)
            self._done = True

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def keys(self):
        self._fill()
        return super().keys()

    def items(self):
        self._fill()
        return super().items()

    def values(self):
        self._fill()
        return super().values()

    def __len__(self):
        self._fill()
        return super().__len__()

    def __contains__(self, k):
        self._fill()
        return super().__contains__(k)
