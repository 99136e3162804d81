"""Placeholder for torchvision (absent here); lets oracle/gen_golden.py import the
reference's model.py:6 and miscc/utils.py:11. Nothing in here is ever executed on the
training-step path. Test infrastructure only."""
from . import utils, models  # noqa: F401
