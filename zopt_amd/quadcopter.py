"""`zopt.quadcopter` under its reference name: the `Quadcopter` class (reference quadcopter.py:8-201) lives in `zopt_amd.models`,
next to the registered device models the kernels evaluate."""
from .models import Quadcopter  # noqa: F401

__all__ = ["Quadcopter"]
