"""Exceptions for non-OK engine statuses.

Same class names and the same status-code mapping as the reference
(graphlearn/python/errors.py): callers catch `gl.OutOfRangeError` etc. unchanged.
"""
from graphlearn import pywrap_graphlearn as pywrap


class OpError(Exception):
  """An operator returned a non-OK graphlearn::Status."""

  def __init__(self, message, error_code=None):
    super(OpError, self).__init__(message)
    self._message = message
    self._error_code = error_code

  @property
  def message(self):
    return self._message

  @property
  def error_code(self):
    return self._error_code

  def __str__(self):
    return self._message


def _make(name):
  return type(name, (OpError,), {"__doc__": "Status code %s." % name})


_CODE_NAMES = [
    ("CANCELLED", "CancelledError"), ("UNKNOWN", "UnknownError"),
    ("INVALID_ARGUMENT", "InvalidArgumentError"), ("DEADLINE_EXCEEDED", "DeadlineExceededError"),
    ("NOT_FOUND", "NotFoundError"), ("ALREADY_EXISTS", "AlreadyExistsError"),
    ("PERMISSION_DENIED", "PermissionDeniedError"), ("UNAUTHENTICATED", "UnauthenticatedError"),
    ("RESOURCE_EXHAUSTED", "ResourceExhaustedError"), ("FAILED_PRECONDITION", "FailedPreconditionError"),
    ("ABORTED", "AbortedError"), ("OUT_OF_RANGE", "OutOfRangeError"),
    ("UNIMPLEMENTED", "UnimplementedError"), ("INTERNAL", "InternalError"),
    ("UNAVAILABLE", "UnavailableError"), ("DATA_LOSS", "DataLossError"),
]
_BY_CODE = {}
for _code, _cls in _CODE_NAMES:
  globals()[_cls] = _make(_cls)
  _BY_CODE[getattr(pywrap.ErrorCode, _code)] = globals()[_cls]

globals()["RequestStopError"] = _make("RequestStopError")  # errors.py:168-171 (a stopped server: code REQUEST_STOP)
_BY_CODE[pywrap.ErrorCode.REQUEST_STOP] = globals()["RequestStopError"]
BaseError = OpError  # the reference's name for the root of these exceptions (errors.py:22-46)
_BY_CLASS = dict((cls, code) for code, cls in _BY_CODE.items())


def exception_type_from_error_code(error_code):
  return _BY_CODE[error_code]


def error_code_from_exception_type(cls):
  return _BY_CLASS[cls]


__all__ = ["OpError", "BaseError", "RequestStopError", "raise_exception_on_not_ok_status", "exception_type_from_error_code",
           "error_code_from_exception_type"] + [c for _, c in _CODE_NAMES]


def raise_exception_on_not_ok_status(status):
  if status.ok():
    return
  cls = _BY_CODE.get(status.code(), UnknownError)  # noqa: F821 -- created above
  raise cls(status.to_string(), status.code())
