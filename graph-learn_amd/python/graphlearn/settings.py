"""Process-wide engine flags (graphlearn/python/config.py; GLOBAL_FLAG in
graphlearn/src/include/config.h).  The thread-pool / RPC knobs of the reference's
service layer are accepted and ignored: there is no service layer between Python and
the GPU here.  `set_sampling_seed` and `set_device_id` are new."""
from graphlearn import pywrap_graphlearn as pywrap

__all__ = [
    "set_default_neighbor_id", "set_padding_mode", "set_default_int_attribute",
    "set_default_float_attribute", "set_default_string_attribute", "set_default_weight",
    "set_default_label", "set_default_timestamp", "set_ignore_invalid", "set_sampler_retry_times",
    "set_default_full_nbr_num",
    "set_inter_threadnum", "set_intra_threadnum", "set_inner_threadnum", "set_datainit_batchsize",
    "set_inmemory_queuesize", "set_shuffle_buffer_size", "set_tracker_mode", "set_storage_mode",
    "set_retry_times", "set_timeout", "set_sampling_seed", "set_device_id",
]


# flags the device-tensor path (NeighborSampler.get_device) passes to the C-ABI itself
_MIRROR = {"padding_mode": 1, "default_neighbor_id": 0, "sampling_seed": 0, "default_float_attr": 0.0, "device_id": 0,
           "default_full_nbr_num": 100, "default_weight": 0.0}


def set_default_neighbor_id(nbr_id):
  pywrap.set_default_neighbor_id(int(nbr_id))
  _MIRROR["default_neighbor_id"] = int(nbr_id)


def set_padding_mode(mode):
  """gl.REPLICATE or gl.CIRCULAR (the engine default, like the reference's)."""
  pywrap.set_padding_mode(int(mode))
  _MIRROR["padding_mode"] = int(mode)


def set_default_int_attribute(value=0):
  pywrap.set_default_int_attr(int(value))


def set_default_float_attribute(value=0.0):
  pywrap.set_default_float_attr(float(value))
  _MIRROR["default_float_attr"] = float(value)


def set_default_string_attribute(value=""):
  pywrap.set_default_string_attr(str(value))


def set_default_weight(value=0.0):
  pywrap.set_default_weight(float(value))
  _MIRROR["default_weight"] = float(value)


def set_default_full_nbr_num(num):
  """Neighbours a node2vec step looks at (GLOBAL_FLAG(DefaultFullNbrNum), 100)."""
  pywrap.set_default_full_nbr_num(int(num))
  _MIRROR["default_full_nbr_num"] = int(num)


def set_default_label(value=-1):
  pywrap.set_default_label(int(value))


def set_default_timestamp(value=-1):
  pywrap.set_default_timestamp(int(value))


def set_ignore_invalid(value):
  pywrap.set_ignore_invalid(1 if value else 0)


def set_sampler_retry_times(value):
  pywrap.set_sampler_retry_times(int(value))


def set_shuffle_buffer_size(value):
  """How many consecutive ids a "shuffle" traversal (node_sampler / edge_sampler / g.V().shuffle(traverse=True)) shuffles at
  a time (GLOBAL_FLAG(ShuffleBufferSize), 10240: node_generator.h:168-190)."""
  pywrap.set_shuffle_buffer_size(int(value))


def set_sampling_seed(seed):
  """Seed of the counter-based random streams of the samplers;
  the reference's samplers cannot be seeded."""
  pywrap.set_sampling_seed(int(seed))
  _MIRROR["sampling_seed"] = int(seed)


def set_device_id(device):
  """GPU that holds this process' graph store (one process per GPU)."""
  pywrap.set_device_id(int(device))
  _MIRROR["device_id"] = int(device)


def _ignored(name):
  def setter(value=0):
    getattr(pywrap, name)(int(value))
  setter.__name__ = name
  setter.__doc__ = "Accepted for source compatibility; has no effect on the device engine."
  return setter


for _n in ("set_inter_threadnum", "set_intra_threadnum", "set_inner_threadnum", "set_datainit_batchsize",
           "set_inmemory_queuesize", "set_tracker_mode", "set_storage_mode",
           "set_retry_times", "set_timeout"):
  globals()[_n] = _ignored(_n)
