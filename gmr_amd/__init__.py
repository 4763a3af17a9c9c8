"""gmr_amd -- MI355X-native batched motion-retargeting engine behind GMR's API.

``from gmr_amd import GeneralMotionRetargeting as GMR`` mirrors
``from general_motion_retargeting import GeneralMotionRetargeting as GMR``.
Heavy imports (torch, the native library) happen on first attribute access.
"""
from .params import IK_CONFIG_DICT, IK_CONFIG_ROOT, ROBOT_BASE_DICT, ROBOT_XML_DICT  # noqa: F401

__all__ = ["GeneralMotionRetargeting", "KinematicsModel", "ROBOT_XML_DICT", "IK_CONFIG_DICT", "ROBOT_BASE_DICT", "IK_CONFIG_ROOT"]


def __getattr__(name):
    if name == "GeneralMotionRetargeting":
        from .motion_retarget import GeneralMotionRetargeting
        return GeneralMotionRetargeting
    if name == "KinematicsModel":
        from .kinematics_model import KinematicsModel
        return KinematicsModel
    raise AttributeError(name)
