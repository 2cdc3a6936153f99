// glx shard communicator (internal interface).  The C-ABI (include/glx.h "shard
// communicator") wraps one of three transports behind this class; glx_dist.hip drives it.
#ifndef GLX_COMM_H_
#define GLX_COMM_H_
#include <stddef.h>
#include <stdint.h>

#include "glx_common.h"

// One buffer pair of an exchange.  All segments of one call share the element counts
// (a request's ids and their row indices travel together, as HashPartitioner copies every
// tensor of a request: hash_partitioner.h:69-74) and go out in ONE transport group.
struct GlxSeg {
  const void* send;
  void* recv;
  size_t elem_bytes;
};

struct glx_comm {
  int device = 0, rank = 0, world = 1, kind = 0;
  int64_t max_message_bytes = (int64_t)512 << 20;
  int64_t last_rounds = 0;
  virtual ~glx_comm() {}
  // Device buffers.  Rank p receives send[send_offs[p] .. + send_counts[p]) of every
  // segment; what rank q sent lands at recv[recv_offs[q] .. + recv_counts[q]).  Offsets and
  // counts are host arrays in elements.  Enqueued on `s` (host-staged transports block).
  virtual int alltoallv(const GlxSeg* segs, int nseg, const int64_t* send_counts, const int64_t* send_offs,
                        const int64_t* recv_counts, const int64_t* recv_offs, hipStream_t s) = 0;
  // d_vals[nvals] (device) of every rank -> h_out[world * nvals] (host), valid on return.
  virtual int allgather_i64(const int64_t* d_vals, int nvals, int64_t* h_out, hipStream_t s) = 0;
  virtual int barrier(hipStream_t s) = 0;
};

#endif  // GLX_COMM_H_
