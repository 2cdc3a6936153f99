// HashPartitioner / Stitch for the host layer (see partition.h).
#include "graphlearn/partition.h"

#include <cfloat>
#include <cstdlib>

#include "graphlearn/aggregating_request.h"
#include "graphlearn/sampling_request.h"

namespace graphlearn {

int32_t HashPartitioner::ShardOf(int64_t id) const { return (int32_t)(std::llabs(id) % range_); }

namespace {
void CopySlice(const Tensor& from, int32_t begin, int32_t end, Tensor* to) {
  switch (from.DType()) {
    case kInt32: to->AddInt32(from.GetInt32() + begin, from.GetInt32() + end); break;
    case kInt64: to->AddInt64(from.GetInt64() + begin, from.GetInt64() + end); break;
    case kFloat: to->AddFloat(from.GetFloat() + begin, from.GetFloat() + end); break;
    case kDouble: to->AddDouble(from.GetDouble() + begin, from.GetDouble() + end); break;
    default: break;
  }
}
}  // namespace

ShardsPtr<OpRequest> HashPartitioner::Partition(const OpRequest* req) const {
  ShardsPtr<OpRequest> ret(new Shards<OpRequest>(range_));
  auto key = req->tensors_.find(req->ShardKey());
  if (!req->IsShardable() || key == req->tensors_.end()) {
    ret->Add(0, const_cast<OpRequest*>(req), false);
    return ret;
  }
  const Tensor& part_by = key->second;
  const int32_t length = part_by.Size();
  for (int32_t index = 0; index < length; ++index) {
    const int32_t shard = ShardOf(part_by.GetInt64(index));
    OpRequest* part = ret->Get(shard);
    if (part == nullptr) {
      part = req->Clone();
      part->DisableShard();
      for (const auto& it : req->tensors_) {
        part->tensors_.erase(it.first);
        ADD_TENSOR(part->tensors_, it.first, it.second.DType(), 16);
      }
      ADD_TENSOR(part->tensors_, kRngRows, kInt64, 16);
      ret->Add(shard, part, true);
    }
    ret->StickerPtr()->Add(shard, index);
    for (const auto& it : req->tensors_) {
      const int32_t dim = it.second.Size() / length;
      CopySlice(it.second, index * dim, (index + 1) * dim, &part->tensors_[it.first]);
    }
    part->tensors_[kRngRows].AddInt64(index);
  }
  return ret;
}

void StitchDense(ShardsPtr<OpResponse> shards, OpResponse* out) {
  int32_t shard_id = 0;
  OpResponse* piece = nullptr;
  int32_t total = 0;
  shards->ResetNext();
  while (shards->Next(&shard_id, &piece)) total += (int32_t)shards->StickerPtr()->At(shard_id).size();
  out->batch_size_ = total;
  shards->ResetNext();
  bool first = true;
  while (shards->Next(&shard_id, &piece)) {
    const std::vector<int32_t>& sticker = shards->StickerPtr()->At(shard_id);
    const int32_t bs = (int32_t)sticker.size();
    if (first) {
      out->params_ = piece->params_;
      for (const auto& it : piece->tensors_) {
        const int32_t dim = bs > 0 ? it.second.Size() / bs : 0;
        out->tensors_.erase(it.first);
        ADD_TENSOR(out->tensors_, it.first, it.second.DType(), total * dim);
        out->tensors_[it.first].Resize(total * dim);
      }
      first = false;
    }
    for (const auto& it : piece->tensors_) {
      if (bs == 0) continue;
      const int32_t dim = it.second.Size() / bs;
      Tensor& to = out->tensors_[it.first];
      for (int32_t i = 0; i < bs; ++i) {
        for (int32_t c = 0; c < dim; ++c) {
          switch (it.second.DType()) {
            case kInt32: to.SetInt32(sticker[i] * dim + c, it.second.GetInt32(i * dim + c)); break;
            case kInt64: to.SetInt64(sticker[i] * dim + c, it.second.GetInt64(i * dim + c)); break;
            case kFloat: to.SetFloat(sticker[i] * dim + c, it.second.GetFloat(i * dim + c)); break;
            default: break;
          }
        }
      }
    }
  }
}

void SamplingResponse::Stitch(ShardsPtr<OpResponse> shards) {
  StitchDense(shards, this);
  const int32_t k = params_.count(kNeighborCount) ? params_[kNeighborCount].GetInt32(0) : 0;
  shape_ = Shape(batch_size_, k);
}

// Combine the shards' partial aggregates.  Shard s contributes segment i only if it
// saw ids of it (count > 0); Mean re-weights partial means by their counts
// (mean_aggregator.cc:31-37,45-61); an overall empty segment is DefaultFloatAttribute.
void AggregatingResponse::Stitch(ShardsPtr<OpResponse> shards, float default_attr) {
  int32_t shard_id = 0;
  OpResponse* tmp = nullptr;
  shards->ResetNext();
  if (!shards->Next(&shard_id, &tmp)) return;
  AggregatingResponse* first = static_cast<AggregatingResponse*>(tmp);
  const std::string name = first->Name();
  const int32_t dim = first->EmbeddingDim();
  const int32_t sg = first->NumSegments();
  SetEmbeddingDim(dim);
  SetNumSegments(sg);
  SetName(name);
  float* emb = MutableEmbeddings();
  int32_t* cnt = MutableSegments();
  enum { kSum, kMean, kMax, kMin, kProd } op = kSum;
  if (name == "MeanAggregator") op = kMean;
  if (name == "MaxAggregator") op = kMax;
  if (name == "MinAggregator") op = kMin;
  if (name == "ProdAggregator") op = kProd;
  // InitFunc (aggregator.cc:61-65, max_/min_/prod_aggregator.cc)
  const float init = op == kMax ? (float)FLT_MIN_10_EXP : op == kMin ? FLT_MAX : op == kProd ? 1.0f : 0.0f;
  for (int32_t i = 0; i < sg; ++i) cnt[i] = 0;
  for (int64_t i = 0; i < (int64_t)sg * dim; ++i) emb[i] = init;
  shards->ResetNext();
  while (shards->Next(&shard_id, &tmp)) {
    AggregatingResponse* part = static_cast<AggregatingResponse*>(tmp);
    const float* pe = part->Embeddings();
    const int32_t* pc = part->Segments();
    for (int32_t i = 0; i < sg; ++i) {
      if (pc[i] == 0) continue;  // this shard holds none of the segment's ids
      float* e = emb + (int64_t)i * dim;
      const float* p = pe + (int64_t)i * dim;
      for (int32_t c = 0; c < dim; ++c) {
        const float v = op == kMean ? p[c] * pc[i] : p[c];  // partial mean -> partial sum
        if (op == kSum || op == kMean) e[c] = e[c] + v;
        else if (op == kMax) e[c] = (e[c] < v) ? v : e[c];
        else if (op == kMin) e[c] = (v < e[c]) ? v : e[c];
        else e[c] = e[c] * v;
      }
      cnt[i] += pc[i];
    }
  }
  for (int32_t i = 0; i < sg; ++i) {
    float* e = emb + (int64_t)i * dim;
    if (cnt[i] == 0) {
      for (int32_t c = 0; c < dim; ++c) e[c] = default_attr;
    } else if (op == kMean) {
      for (int32_t c = 0; c < dim; ++c) e[c] = e[c] / cnt[i];
    }
  }
}

}  // namespace graphlearn
