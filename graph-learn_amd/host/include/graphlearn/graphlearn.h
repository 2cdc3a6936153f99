// Umbrella header of the glx host layer (C++ mirror of the graphlearn::op API).
#ifndef GLX_HOST_GRAPHLEARN_H_
#define GLX_HOST_GRAPHLEARN_H_
#include "graphlearn/aggregating_request.h"
#include "graphlearn/client.h"
#include "graphlearn/config.h"
#include "graphlearn/dag.h"
#include "graphlearn/data_source.h"
#include "graphlearn/graph_request.h"
#include "graphlearn/graph_store.h"
#include "graphlearn/op_request.h"
#include "graphlearn/operator.h"
#include "graphlearn/op_runner.h"
#include "graphlearn/partition.h"
#include "graphlearn/sampling_request.h"
#include "graphlearn/status.h"
#include "graphlearn/subgraph_request.h"
#include "graphlearn/tensor.h"
#endif
