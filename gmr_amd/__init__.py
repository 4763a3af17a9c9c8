"""gmr_amd -- MI355X-native batched motion-retargeting engine behind GMR's API."""
from .params import IK_CONFIG_DICT, IK_CONFIG_ROOT, ROBOT_BASE_DICT, ROBOT_XML_DICT  # noqa: F401
