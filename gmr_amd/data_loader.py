"""``general_motion_retargeting.data_loader`` (data_loader.py:4-18): the reader of the motion files the dataset scripts write."""
from .dataset import load_robot_motion  # noqa: F401

__all__ = ["load_robot_motion"]
