"""gmr_amd -- MI355X-native batched motion-retargeting engine behind GMR's API.

``from gmr_amd import GeneralMotionRetargeting as GMR`` mirrors
``from general_motion_retargeting import GeneralMotionRetargeting as GMR``.
Heavy imports (torch, the native library) happen on first attribute access.
"""
from .params import ASSET_ROOT, IK_CONFIG_DICT, IK_CONFIG_ROOT, ROBOT_BASE_DICT, ROBOT_XML_DICT, VIEWER_CAM_DISTANCE_DICT  # noqa: F401

# the names general_motion_retargeting/__init__.py:2-6 exports (RobotMotionViewer is out of scope: asking for it says so)
__all__ = ["GeneralMotionRetargeting", "KinematicsModel", "load_robot_motion", "ROBOT_XML_DICT", "IK_CONFIG_DICT", "ROBOT_BASE_DICT", "IK_CONFIG_ROOT",
           "ASSET_ROOT", "VIEWER_CAM_DISTANCE_DICT"]


def __getattr__(name):
    if name == "GeneralMotionRetargeting":
        from .motion_retarget import GeneralMotionRetargeting
        return GeneralMotionRetargeting
    if name == "KinematicsModel":
        from .kinematics_model import KinematicsModel
        return KinematicsModel
    if name == "load_robot_motion":
        from .dataset import load_robot_motion
        return load_robot_motion
    if name == "RobotMotionViewer":
        raise AttributeError("gmr_amd has no RobotMotionViewer: the MuJoCo viewer is outside this engine's scope (use the reference's with the qpos this engine returns)")
    raise AttributeError(name)
