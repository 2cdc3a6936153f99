// Request partitioning and response stitching for the C++ host layer: the in-process
// equivalent of the reference's distributed fan-out (core/runner/op_runner.h:60-152):
//   HashPartitioner<T>::Partition  -- core/partition/hash_partitioner.h:33-92
//                                     (shard = llabs(id) % range, stable inside a shard,
//                                     Sticker = original indices, include/shardable.h:26-59)
//   OpResponse::Stitch             -- core/partition/stitcher.h:31-108 (dense tensors)
//   AggregatingResponse::Stitch    -- service/request/aggregating_request.cc:172-213
// A caller (e.g. a multi-GPU server with one GraphStore per device) partitions a
// request, runs Process() per shard and stitches.  Unlike the reference, every part
// request also carries the original row indices (kRngRows) so that the samplers draw
// from the same random streams as the unpartitioned request,
// and the aggregate stitch ignores shards that saw no id of a segment, so
// Max/Min/Prod equal the single-shard result (the reference folds the empty shards'
// default value in: SURVEY.md 8(a) quirk 8).  Sparse (FullSampler) responses are not
// stitched here.
#ifndef GLX_HOST_PARTITION_H_
#define GLX_HOST_PARTITION_H_
#include <cstdint>
#include <memory>
#include <vector>

#include "graphlearn/op_request.h"

namespace graphlearn {

class Sticker {
public:
  explicit Sticker(int32_t capacity) : values_(capacity) {}
  void Add(int32_t shard_id, int32_t original_index) { values_[shard_id].push_back(original_index); }
  const std::vector<int32_t>& At(int32_t shard_id) const { return values_[shard_id]; }
  int32_t Capacity() const { return (int32_t)values_.size(); }

private:
  std::vector<std::vector<int32_t>> values_;
};

template <class T>
class Shards {
public:
  explicit Shards(int32_t capacity) : pieces_(capacity, nullptr), owned_(capacity, false), sticker_(capacity), cursor_(0) {}
  ~Shards() {
    for (size_t i = 0; i < pieces_.size(); ++i) {
      if (owned_[i]) delete pieces_[i];
    }
  }
  void Add(int32_t shard_id, T* t, bool own) {
    pieces_[shard_id] = t;
    owned_[shard_id] = own;
  }
  T* Get(int32_t shard_id) const { return shard_id < Capacity() ? pieces_[shard_id] : nullptr; }
  int32_t Capacity() const { return (int32_t)pieces_.size(); }
  int32_t Size() const {
    int32_t n = 0;
    for (T* p : pieces_) n += p != nullptr;
    return n;
  }
  Sticker* StickerPtr() { return &sticker_; }
  // Iterate over the non-empty shards in shard order.
  bool Next(int32_t* shard_id, T** t) {
    while (cursor_ < Capacity() && pieces_[cursor_] == nullptr) ++cursor_;
    if (cursor_ >= Capacity()) return false;
    *shard_id = cursor_;
    *t = pieces_[cursor_++];
    return true;
  }
  void ResetNext() { cursor_ = 0; }

private:
  std::vector<T*> pieces_;
  std::vector<bool> owned_;
  Sticker sticker_;
  int32_t cursor_;
};

template <class T>
using ShardsPtr = std::shared_ptr<Shards<T>>;

class HashPartitioner {
public:
  explicit HashPartitioner(int32_t range) : range_(range) {}
  // Splits `req` by its shard-key tensor (kSrcIds / kNodeIds).  Every per-element
  // tensor of the request follows its element; the part requests are Clone()s.
  ShardsPtr<OpRequest> Partition(const OpRequest* req) const;
  int32_t ShardOf(int64_t id) const;

private:
  int32_t range_;
};

// Dense stitch: row i of shard s goes to row Sticker(s)[i] (stitcher.h:82-93).
void StitchDense(ShardsPtr<OpResponse> shards, OpResponse* out);

}  // namespace graphlearn
#endif  // GLX_HOST_PARTITION_H_
