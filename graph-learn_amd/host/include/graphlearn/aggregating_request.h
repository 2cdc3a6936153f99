// AggregatingRequest / AggregatingResponse with the reference's surface
// (graphlearn/src/include/aggregating_request.h:24-91).
#ifndef GLX_HOST_AGGREGATING_REQUEST_H_
#define GLX_HOST_AGGREGATING_REQUEST_H_
#include <string>

#include "graphlearn/op_request.h"
#include "graphlearn/partition.h"

namespace graphlearn {

// Request: node ids + their (non-decreasing) segment ids + the segment count.
class AggregatingRequest : public OpRequest {
public:
  AggregatingRequest();
  AggregatingRequest(const std::string& type, const std::string& strategy);

  void Set(const int64_t* node_ids, const int32_t* segment_ids, int32_t num_ids, int32_t num_segments);
  OpRequest* Clone() const override;

  const std::string& Type() const;      // node type
  const std::string& Strategy() const;  // operator name, e.g. "SumAggregator"
  int32_t NumIds() const;
  int32_t NumSegments() const { return num_segments_; }

  // The reference's cursor (aggregating_request.cc:86-105), kept for callers that walk the
  // request; the device path reads the two arrays directly.
  bool Next(int64_t* node_id, int32_t* segment_id);
  bool SegmentEnd(int32_t segment_id) const;
  const int64_t* NodeIds() const;
  const int32_t* SegmentIds() const;

private:
  int32_t cursor_;
  int32_t num_segments_;
};

// Response: [NumSegments, EmbeddingDim] floats + the id count of every segment.
class AggregatingResponse : public OpResponse {
public:
  AggregatingResponse();
  OpResponse* New() const override { return new AggregatingResponse; }
  void Swap(OpResponse& right) override;

  void SetName(const std::string& name);
  void SetEmbeddingDim(int32_t dim);
  void SetNumSegments(int32_t num_segments);
  void AppendEmbedding(const float* value);
  void AppendSegment(int32_t size);

  std::string Name() const { return name_; }
  int32_t EmbeddingDim() const { return emb_dim_; }
  int32_t NumSegments() const { return batch_size_; }
  const float* Embeddings() const;
  const int32_t* Segments() const;

  // Combine per-shard partial aggregates (aggregating_request.cc:172-213); see partition.h.
  void Stitch(ShardsPtr<OpResponse> shards, float default_attr = 0.0f);

  // Device-path additions: size the outputs once, fill them with one copy.
  float* MutableEmbeddings();
  int32_t* MutableSegments();

private:
  std::string name_;
  int32_t emb_dim_;
};

}  // namespace graphlearn
#endif  // GLX_HOST_AGGREGATING_REQUEST_H_
