"""NodeState / EdgeState (graphlearn/python/data/state.py): thread-safe counters keyed by type, which the reference's
traversal samplers keep their client-side cursors in."""
import threading


class State(object):

  def __init__(self, value=None):
    self._values = {} if value is None else value
    self._value_lock = threading.Lock()

  def put(self, key, value):
    with self._value_lock:
      self._values[key] = value

  def get(self, key):
    return self._values.get(key, 0)

  def inc(self, key, delta=1):
    with self._value_lock:
      self._values[key] = self._values.get(key, 0) + delta


class NodeState(State):
  pass


class EdgeState(State):
  pass


class DagState(State):
  pass
