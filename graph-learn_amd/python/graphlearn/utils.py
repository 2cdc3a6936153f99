"""Small helpers shared by the Python layer (graphlearn/python/utils.py)."""
from enum import Enum

__all__ = ["Mask", "get_mask_type", "strategy2op"]


def strategy2op(strategy, op_type):
  """"random_without_replacement", "Sampler" -> "RandomWithoutReplacementSampler"."""
  words = strategy.split("_") if isinstance(strategy, str) else []
  return "".join(w.capitalize() for w in words) + op_type


class Mask(Enum):
  NONE = 0
  TRAIN = 1
  TEST = 2
  VAL = 3


def get_mask_type(raw_type, mask=Mask.NONE):
  """Storage type name of a masked source: "user" + TRAIN/TEST/VAL ->
  "MASK*user" / "MASK**user" / "MASK***user" (one star per mask value)."""
  if not isinstance(raw_type, str):
    raise ValueError("type must be a string")
  if isinstance(mask, str):
    mask = Mask[mask.upper()]
  if not isinstance(mask, Mask):
    raise ValueError("mask must be a Mask")
  return raw_type if mask is Mask.NONE else "MASK" + "*" * mask.value + raw_type
