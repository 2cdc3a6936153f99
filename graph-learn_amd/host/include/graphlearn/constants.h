// Tensor / parameter keys (graphlearn/src/include/constants.h:22-74).  The VALUES are the
// reference's too (service/constants.cc:20-72): its Python layer spells some of them out
// ("ia", "fa", "wei", "ts", "filt", "dg" in python/gsl/dag_dataset.py:137-152, dag_node.py:329-331),
// so they are part of the interface, not an implementation detail.
#ifndef GLX_HOST_CONSTANTS_H_
#define GLX_HOST_CONSTANTS_H_
namespace graphlearn {
extern const char* kUnspecified;
extern const char* kOpName;
extern const char* kNodeType;
extern const char* kEdgeType;
extern const char* kType;
extern const char* kSrcType;
extern const char* kDstType;
extern const char* kSrcIds;
extern const char* kDstIds;
extern const char* kNodeIds;
extern const char* kEdgeIds;
extern const char* kNeighborCount;
extern const char* kNeighborIds;
extern const char* kBatchSize;
extern const char* kIsSparse;
extern const char* kStrategy;
extern const char* kDegreeKey;
extern const char* kWeightKey;
extern const char* kLabelKey;
extern const char* kTimestampKey;
extern const char* kIntAttrKey;
extern const char* kFloatAttrKey;
extern const char* kStringAttrKey;
extern const char* kSideInfo;
extern const char* kDirection;
extern const char* kSegmentIds;
extern const char* kNumSegments;
extern const char* kSegments;
extern const char* kDistances;
extern const char* kRowIndices;
extern const char* kColIndices;
extern const char* kSeedType;
extern const char* kNbrType;
extern const char* kCount;
extern const char* kBatchShare;
extern const char* kUnique;
extern const char* kIntCols;
extern const char* kIntProps;
extern const char* kFloatCols;
extern const char* kFloatProps;
extern const char* kStrCols;
extern const char* kStrProps;
extern const char* kFilterType;
extern const char* kFilterField;
extern const char* kFilterValues;
extern const char* kDegrees;
extern const char* kEpoch;
extern const char* kNodeFrom;
extern const char* kNeedDist;
extern const char* kDistToSrc;
extern const char* kDistToDst;
extern const char* kSparseIds;
// glx additions (no counterpart in the reference): the pinned random stream of a request, and the
// original row indices of a part of a partitioned request.
extern const char* kCallCounter;
extern const char* kRngRows;
}  // namespace graphlearn
#endif  // GLX_HOST_CONSTANTS_H_
