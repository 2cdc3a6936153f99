// Tensor / parameter keys: ONE table of (name, wire string) that declares them here, defines them in base.cc and exports
// them to Python (pywrap_graphlearn.cc).  Names as in graphlearn/src/include/constants.h:22-74; the STRINGS are the
// reference's too (service/constants.cc:20-72): its Python layer spells some of them out ("ia", "fa", "wei", "ts",
// "filt", "dg" in python/gsl/dag_dataset.py:137-152, dag_node.py:329-331), so they are part of the interface, not an
// implementation detail.  The last two are glx additions without a counterpart: the pinned random stream of a request
// and the original row indices of a part of a partitioned request.
#ifndef GLX_HOST_CONSTANTS_H_
#define GLX_HOST_CONSTANTS_H_
#define GLX_TENSOR_KEYS(X) \
  X(kUnspecified, "unspecified") X(kOpName, "op") X(kNodeType, "nt") X(kEdgeType, "et") X(kType, "tp") \
  X(kSrcType, "st") X(kDstType, "dt") X(kSrcIds, "sid") X(kDstIds, "did") X(kNodeIds, "nid") X(kEdgeIds, "eid") \
  X(kNeighborCount, "nbc") X(kNeighborIds, "nbi") X(kBatchSize, "bs") X(kIsSparse, "is") X(kStrategy, "str") \
  X(kDegreeKey, "deg") X(kWeightKey, "wei") X(kLabelKey, "lb") X(kTimestampKey, "ts") X(kIntAttrKey, "ia") \
  X(kFloatAttrKey, "fa") X(kStringAttrKey, "sa") X(kSideInfo, "si") X(kDirection, "dir") X(kSegmentIds, "segi") \
  X(kNumSegments, "ns") X(kSegments, "sm") X(kDistances, "dis") X(kRowIndices, "ridx") X(kColIndices, "cidx") \
  X(kSeedType, "seedt") X(kNbrType, "nbrt") X(kCount, "cnt") X(kBatchShare, "batch_share") X(kUnique, "unique") \
  X(kIntCols, "icols") X(kIntProps, "ipps") X(kFloatCols, "fcols") X(kFloatProps, "fpps") X(kStrCols, "scols") \
  X(kStrProps, "spps") X(kFilterType, "ftype") X(kFilterField, "field") X(kFilterValues, "filt") \
  X(kDegrees, "dg") X(kEpoch, "ep") X(kNodeFrom, "nf") X(kNeedDist, "need_dist") X(kDistToSrc, "dist_to_src") \
  X(kDistToDst, "dist_to_dst") X(kSparseIds, "sparse_ids") X(kCallCounter, "call_counter") \
  X(kRngRows, "rng_rows")
namespace graphlearn {
#define GLX_DECLARE_TENSOR_KEY(name, wire) extern const char* name;
GLX_TENSOR_KEYS(GLX_DECLARE_TENSOR_KEY)
#undef GLX_DECLARE_TENSOR_KEY
}  // namespace graphlearn
#endif  // GLX_HOST_CONSTANTS_H_
