"""``general_motion_retargeting.utils.lafan1`` (utils/lafan1.py:8-71) on this engine."""
from __future__ import annotations


def load_lafan1_file(bvh_file):
    """-> (frames, human_height): ``frames[t] = {bone: (position[3] in metres, Z-up; quaternion wxyz[4])}`` incl. the LeftFootMod / RightFootMod
    entries, and the height estimate of the last frame -- the reference's return value, computed by the native parser and ``gmr_bvh_fk``.
    (``gmr_amd.bvh.load_lafan1_file`` returns the same numbers as GPU tensors for ``retarget_batch``.)"""
    from ..bvh import load_lafan1_file as _load
    clip = _load(str(bvh_file))
    return clip.frames(), clip.human_height
