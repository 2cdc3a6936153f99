// Drop-in path benchmark: T host threads issue independent 2-hop sampling +
// aggregation requests THROUGH THE C++ OPERATOR API (OpFactory::Create(name)->
// Process(req, res), host buffers) -- the reference's own concurrency model (one
// request per pool thread, in_memory_service.cc:64-71) and the same request shape
// bench.py's cpu_baseline times on the reference.  Everything crosses PCIe, so this
// is the PCIe-inclusive rate of the drop-in boundary, not the device-resident rate.
//
//   host_path_bench [threads=8] [seeds_per_request=1024] [requests_per_thread=20]
//                   [log2_nodes=21] [edges=20000000] [dim=256]
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
  const int64_t V = 1LL << scale;
  const int k1 = 25, k2 = 10;

  // RMAT (0.57, 0.19, 0.19, 0.05) edge stream with U(0.01, 1) weights
  std::mt19937_64 rng(4);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
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
      for (int l = 0; l < scale; ++l) {
        const double r = uni(rng);
        s = (s << 1) | (r >= 0.76);
        d = (d << 1) | ((r >= 0.57 && r < 0.76) || r >= 0.95);
      }
      v.src_id = s;
      v.dst_id = d;
      v.weight = static_cast<float>(0.01 + 0.99 * uni(rng));
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
    std::mt19937 frng(5);
    std::uniform_real_distribution<float> fu(-1.f, 1.f);
    for (int64_t i = 0; i < V; ++i) {
      nv.id = i;
      for (int j = 0; j < D; ++j) nv.attrs[j] = fu(frng);
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
