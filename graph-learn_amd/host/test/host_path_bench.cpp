// Drop-in path benchmark: T host threads issue independent 2-hop sampling +
// aggregation requests THROUGH THE C++ OPERATOR API (OpFactory::Create(name)->
// Process(req, res), host buffers) -- the reference's own concurrency model (one
// request per pool thread, in_memory_service.cc:64-71) and the same request shape
// bench.py's cpu_baseline times on the reference.  Everything crosses PCIe, so this
// is the PCIe-inclusive rate of the drop-in boundary, not the device-resident rate.
//
//   host_path_bench [threads=8] [seeds_per_request=1024] [requests_per_thread=20]
//                   [log2_nodes=21] [edges=20000000] [dim=256] [nodes=2^log2_nodes]
// bench.py runs it on the headline graph's shape: log2_nodes 24, edges 100000000, dim 256, nodes 10000000
// (RMAT ids folded onto [0, nodes) like synth.rmat_edges_torch does).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "graphlearn/graphlearn.h"

using namespace graphlearn;      // NOLINT
using namespace graphlearn::op;  // NOLINT

int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 8;
  const int B = argc > 2 ? atoi(argv[2]) : 1024;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  const int scale = argc > 4 ? atoi(argv[4]) : 21;
  const int64_t E = argc > 5 ? atoll(argv[5]) : 20000000;
  const int D = argc > 6 ? atoi(argv[6]) : 256;
  const int64_t V = argc > 7 && atoll(argv[7]) > 0 ? atoll(argv[7]) : 1LL << scale;
  const int k1 = 25, k2 = 10;

  // RMAT (0.57, 0.19, 0.19, 0.05) edge stream with U(0.01, 1) weights; splitmix64, 16 random bits per level
  uint64_t sm = 4;
  auto next64 = [&sm]() {
    uint64_t z = (sm += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  GraphStore store;
  {
    io::SideInfo info;
    info.format = io::kWeighted;
    info.type = "e";
    Graph* graph = store.GetGraph("e");
    graph->SetSideInfo(&info);
    io::EdgeValue v;
    for (int64_t e = 0; e < E; ++e) {
      int64_t s = 0, d = 0;
      uint64_t bits = 0;
      for (int l = 0; l < scale; ++l) {
        if ((l & 3) == 0) bits = next64();
        const uint32_t r = (uint32_t)(bits & 0xFFFF);  // U[0, 65536): thresholds 0.57 / 0.76 / 0.95
        bits >>= 16;
        s = (s << 1) | (r >= 49807);
        d = (d << 1) | ((r >= 37356 && r < 49807) || r >= 62259);
      }
      v.src_id = s % V;
      v.dst_id = d % V;
      v.weight = 0.01f + 0.99f * (float)((next64() >> 40) * (1.0 / 16777216.0));
      graph->Add(&v);
    }
    IndexOption opt;
    opt.name = "sort";
    Status st = graph->Build(opt);
    if (!st.ok()) { std::printf("graph build failed: %s\n", st.ToString().c_str()); return 2; }
  }
  {
    io::SideInfo info;
    info.format = io::kAttributed;
    info.f_num = D;
    info.type = "n";
    Noder* noder = store.GetNoder("n");
    noder->SetSideInfo(&info);
    io::NodeValue nv;
    nv.attrs.resize(D);
    sm = 5;
    for (int64_t i = 0; i < V; ++i) {
      nv.id = i;
      for (int j = 0; j + 1 < D; j += 2) {
        const uint64_t z = next64();
        nv.attrs[j] = (float)(uint32_t)(z >> 40) * (2.0f / 16777216.0f) - 1.0f;
        nv.attrs[j + 1] = (float)(uint32_t)((z >> 8) & 0xFFFFFF) * (2.0f / 16777216.0f) - 1.0f;
      }
      if (D & 1) nv.attrs[D - 1] = (float)(uint32_t)(next64() >> 40) * (2.0f / 16777216.0f) - 1.0f;
      noder->Add(&nv);
    }
    IndexOption opt;
    Status st = noder->Build(opt);
    if (!st.ok()) { std::printf("noder build failed: %s\n", st.ToString().c_str()); return 2; }
  }
  OpFactory::GetInstance()->Set(&store);
  Operator* sampler = OpFactory::GetInstance()->Create("EdgeWeightSampler");
  Operator* agg = OpFactory::GetInstance()->Create("MaxAggregator");

  std::vector<int32_t> seg2((size_t)B * k1 * k2), seg1((size_t)B * k1);
  for (size_t i = 0; i < seg2.size(); ++i) seg2[i] = (int32_t)(i / k2);
  for (size_t i = 0; i < seg1.size(); ++i) seg1[i] = (int32_t)(i / k1);

  auto worker = [&](int t, int n, int64_t* edges, int* ok) {
    std::mt19937_64 r(100 + t);
    std::vector<int64_t> seeds(B);
    for (int rep = 0; rep < n; ++rep) {
      for (auto& s : seeds) s = (int64_t)(r() % (uint64_t)V);
      SamplingRequest q1("e", "EdgeWeightSampler", k1);
      SamplingResponse r1;
      q1.Set(seeds.data(), B);
      if (!sampler->Process(&q1, &r1).ok()) { *ok = 0; return; }
      SamplingRequest q2("e", "EdgeWeightSampler", k2);
      SamplingResponse r2;
      q2.Set(r1.GetNeighborIds(), B * k1);
      if (!sampler->Process(&q2, &r2).ok()) { *ok = 0; return; }
      AggregatingRequest a2("n", "MaxAggregator");
      AggregatingResponse o2;
      a2.Set(r2.GetNeighborIds(), seg2.data(), B * k1 * k2, B * k1);
      if (!agg->Process(&a2, &o2).ok()) { *ok = 0; return; }
      AggregatingRequest a1("n", "MaxAggregator");
      AggregatingResponse o1;
      a1.Set(r1.GetNeighborIds(), seg1.data(), B * k1, B);
      if (!agg->Process(&a1, &o1).ok()) { *ok = 0; return; }
      *edges += (int64_t)B * k1 + (int64_t)B * k1 * k2;
    }
  };
  // A persistent pool, like the reference's (common/threading/runner/threadpool.h): every thread first
  // serves two warm-up requests (its device workspaces and streams are per thread; the response blocks are
  // pinned once, when BlockPool first obtains them, and recycled afterwards), then all start the timed
  // requests together.
  std::vector<std::thread> pool;
  std::vector<int64_t> edges(threads, 0), warm_edges(threads, 0);
  std::vector<int> oks(threads, 1);
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::chrono::steady_clock::time_point t0;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t] {
      worker(t, 2, &warm_edges[t], &oks[t]);
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      if (oks[t]) worker(t, reps, &edges[t], &oks[t]);
    });
  }
  while (ready.load() < threads) std::this_thread::yield();
  t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& th : pool) th.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int64_t tot = 0;
  for (int t = 0; t < threads; ++t) { tot += edges[t]; if (!oks[t]) { std::printf("a request failed\n"); return 2; } }
  std::printf("{\"threads\": %d, \"seeds_per_request\": %d, \"requests_per_thread\": %d, \"nodes\": %lld, \"edges\": %lld, "
              "\"dim\": %d, \"wall_s\": %.4f, \"sampled_edges_per_s_host_pointer_path\": %.4g}\n",
              threads, B, reps, (long long)V, (long long)E, D, dt, tot / dt);
  return 0;
}
