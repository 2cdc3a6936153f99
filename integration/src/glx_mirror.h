// INTEGRATION.md 1.1 -- the device mirror of a reference GraphStore.
//
// This directory is what a maintainer of alibaba/graph-learn ADDS to the reference tree to route the sampling /
// aggregation operators through libglx.so.  It is compiled HERE against the reference's own headers and linked with the
// reference's own registry / factory / request / storage translation units (integration/Makefile), so the drop-in is
// proven inside graphlearn::op itself, not only inside this repo's mirror of it.  Test infrastructure like oracle/_ref:
// nothing under graph-learn_amd/ includes or links it.
#ifndef GLX_INTEGRATION_MIRROR_H_
#define GLX_INTEGRATION_MIRROR_H_

#include <string>

#include "core/graph/graph_store.h"
#include "glx.h"

namespace graphlearn {
namespace op {

// Device handles of one GraphStore, made on first use after Build() (graph_store.cc:252-276 fixes the adjacency
// order) and remade when the storage has grown since.  Thread-safe: Process() runs on up to 32 pool threads.
// Return the C-ABI's code (GLX_OK, or why the mirror could not be made: GLX_UNAVAILABLE without a GPU, ...).
int GlxGraphOf(GraphStore* store, const std::string& edge_type, const glx_graph** out);
int GlxFeaturesOf(GraphStore* store, const std::string& node_type, const glx_features** out);

// Status of a failed C-ABI call: the code IS a graphlearn::error::Code, the text is glx_last_error().
Status GlxStatus(int rc);

}  // namespace op
}  // namespace graphlearn

#endif  // GLX_INTEGRATION_MIRROR_H_
