// ORACLE BUILD STUB (test infrastructure).  Stands in for the reference's platform/env.h (thread pools,
// file systems, the naming engine): SubGraphSampler only passes Env::Default() on to GetOpRunner
// (subgraph_sampler.cc:29-31), and the runner stub beside this file defines that Env.
#ifndef GLX_REF_STUB_PLATFORM_ENV_H_
#define GLX_REF_STUB_PLATFORM_ENV_H_
#include "core/runner/op_runner.h"
#endif  // GLX_REF_STUB_PLATFORM_ENV_H_
