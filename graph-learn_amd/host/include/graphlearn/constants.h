// Tensor / parameter keys of the hot path (graphlearn/src/include/constants.h:22-72).
#ifndef GLX_HOST_CONSTANTS_H_
#define GLX_HOST_CONSTANTS_H_
namespace graphlearn {
extern const char* kUnspecified;
extern const char* kOpName;
extern const char* kNodeType;
extern const char* kEdgeType;
extern const char* kType;
extern const char* kSrcIds;
extern const char* kNodeIds;
extern const char* kEdgeIds;
extern const char* kNeighborCount;
extern const char* kStrategy;
extern const char* kFloatAttrKey;
extern const char* kIntAttrKey;
extern const char* kWeightKey;
extern const char* kLabelKey;
extern const char* kTimestampKey;
extern const char* kDegreeKey;
extern const char* kSideInfo;
extern const char* kSegmentIds;
extern const char* kNumSegments;
extern const char* kSegments;
extern const char* kFilterType;
extern const char* kFilterField;
extern const char* kFilterValues;
}  // namespace graphlearn
#endif  // GLX_HOST_CONSTANTS_H_
