// Global flags of the hot path.  Mirrors graphlearn/src/include/config.h:25-162
// (GLOBAL_FLAG(name), SetGlobalFlag<Name>) with the defaults of
// graphlearn/src/common/base/config.cc:77-163 for the flags the samplers and
// aggregators read, plus the two flags the device path adds.
#ifndef GLX_HOST_CONFIG_H_
#define GLX_HOST_CONFIG_H_
#include <cstdint>
#include <string>

namespace graphlearn {
#define GLOBAL_FLAG(name) ::graphlearn::g##name

extern int32_t gPaddingMode;            // config.cc:94  (1 = circular)
extern int64_t gDefaultNeighborId;      // config.cc:98  (0)
extern float gDefaultFloatAttribute;    // config.cc:100 (0.0)
extern float gDefaultWeight;            // config.cc:102 (0.0)
extern int64_t gDefaultIntAttribute;    // config.cc:99  (0)
extern std::string gDefaultStringAttribute;  // config.cc:101 ("")
extern int64_t gDefaultLabel;           // config.cc:103 (-1)
extern int64_t gDefaultTimestamp;       // config.cc:104 (-1)
extern int32_t gIgnoreInvalid;          // config.cc:109 (1: skip records that fail to parse)
extern int32_t gDefaultFullNbrNum;       // config.cc:111 (100: neighbours a node2vec step looks at)
extern int32_t gSamplingRetryTimes;     // config.cc:108 (5; RandomSampler's redraws of filtered neighbours)
extern int32_t gShuffleBufferSize;      // config.cc:88 (10240: a "shuffle" traversal shuffles this many consecutive ids at a time)
// Flags of the layers around the path.  The DAG runner and the dataset read TapeCapacity / DatasetCapacity /
// Timeout / ClientId / ClientCount (core/dag/tape.cc:85-93, core/dag/dag_dataset.cc:28-35,63-66); DeployMode is
// recorded and checked (only kLocal is served in-process); the rest parameterise the reference's RPC service,
// thread pools, storage layout, KNN, vineyard and actor engine -- none of which exists here -- and are kept so that
// the reference's Python layer can set them (python/c/py_export.cc:38-80): stored, never read.
extern int32_t gDeployMode;        // config.cc:77  (0 = local)
extern int32_t gClientId;          // :78
extern int32_t gClientCount;       // :79 (1)
extern int32_t gServerCount;       // :81 (1)
extern int32_t gTimeout;           // :82 (60 s: how long Dataset::Next waits for a batch before it logs and waits on)
extern int32_t gTapeCapacity;      // :85 (10: rounds a query runs ahead of its consumer)
extern int32_t gDatasetCapacity;   // :86 (10: responses a Dataset prefetches)
extern int32_t gTrackerMode;       // :95 (1 = file system)
// New (the reference has no seed flag, include/config.h:77-118): the seed of the
// glx seeding contract, and the GPU this process' GraphStore lives on.
extern int64_t gSamplingSeed;
extern int32_t gDeviceId;

void SetGlobalFlagPaddingMode(int32_t v);
void SetGlobalFlagDefaultNeighborId(int64_t v);
void SetGlobalFlagDefaultFloatAttribute(float v);
void SetGlobalFlagDefaultWeight(float v);
void SetGlobalFlagDefaultIntAttribute(int64_t v);
void SetGlobalFlagDefaultStringAttribute(const std::string& v);
void SetGlobalFlagDefaultLabel(int64_t v);
void SetGlobalFlagDefaultTimestamp(int64_t v);
void SetGlobalFlagIgnoreInvalid(int32_t v);
void SetGlobalFlagSamplingRetryTimes(int32_t v);
void SetGlobalFlagDefaultFullNbrNum(int32_t v);
void SetGlobalFlagShuffleBufferSize(int32_t v);
void SetGlobalFlagDeployMode(int32_t v);
void SetGlobalFlagClientId(int32_t v);
void SetGlobalFlagClientCount(int32_t v);
void SetGlobalFlagServerCount(int32_t v);
void SetGlobalFlagTimeout(int32_t v);
void SetGlobalFlagTapeCapacity(int32_t v);
void SetGlobalFlagDatasetCapacity(int32_t v);
void SetGlobalFlagTrackerMode(int32_t v);
int32_t GetGlobalFlagTrackerMode();
// Stored-only flags: name -> last value set (see above).  Typed by what py_export.cc binds.
void SetGlobalFlagUnused(const char* name, int64_t v);
void SetGlobalFlagUnused(const char* name, const std::string& v);
void SetGlobalFlagSamplingSeed(int64_t v);
void SetGlobalFlagDeviceId(int32_t v);

enum PaddingMode { kReplicate = 0, kCircular = 1 };  // include/constants.h:119-122
enum DeployMode { kLocal = 0, kServer = 1, kWorker = 2 };  // :109-113
enum PartitionMode { kNoPartition = 0, kByHash = 1 };      // :114-117
enum TrackerMode { kRpc = 0, kFileSystem = 1 };            // :124-127
enum NodeFrom { kEdgeSrc = 0, kEdgeDst = 1, kNode = 2 };  // include/constants.h
}  // namespace graphlearn
#endif  // GLX_HOST_CONFIG_H_
