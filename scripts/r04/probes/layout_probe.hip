// Does the DRAM-bound leg of the row gather depend on WHERE the 1 KiB of a row sits?  (scripts/r04/region_sweep.py: ids
// uniform over >= 4 GB run at 5.3 TB/s, over <= 512 MB at 7.7 -- with no difference between dense and spread rows, so
// not address translation: DRAM page / channel locality.)  This probe gathers N random "rows" of 1 KiB from a 10 GiB
// table whose rows are stored as 1024 / piece pieces, piece j of row i of a block of NB rows at
// block + (j * NB + i) * piece -- NB = 1 is the contiguous layout the product uses.  One wave per visit group, U
// visits in flight, lane l reads 16 bytes; accumulates with max and writes one row per wave (keeps the loads alive).
// Not part of the product: a measurement only.   hipcc --offload-arch=gfx950 -O3 layout_probe.hip -o layout_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

template <int U>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ table, uint64_t rows, uint32_t piece_log2,
                                              uint32_t nb_log2, uint32_t visits_per_wave, uint64_t salt,
                                              f4* __restrict__ out) {
  const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t byte = lane * 16;                      // this lane's 16 bytes of the 1 KiB row
  const uint32_t j = byte >> piece_log2;                // piece of the row
  const uint32_t within = byte & ((1u << piece_log2) - 1);
  const uint64_t piece_stride = (uint64_t)1 << (piece_log2 + nb_log2);  // NB * piece
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (uint32_t v = 0; v < visits_per_wave; v += U) {
    f4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t r = __umul64hi(mix(salt + wave * visits_per_wave + v + u), rows);
      const uint64_t blk = r >> nb_log2, i = r & (((uint64_t)1 << nb_log2) - 1);
      const char* p = table + (blk << (10 + nb_log2)) + j * piece_stride + (i << piece_log2) + within;
      x[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc.x = fmaxf(acc.x, x[u].x); acc.y = fmaxf(acc.y, x[u].y); acc.z = fmaxf(acc.z, x[u].z); acc.w = fmaxf(acc.w, x[u].w);
    }
  }
  out[wave * 64 + lane] = acc;
}

int main(int argc, char** argv) {
  const uint64_t table_rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;  // 1 KiB each
  const uint64_t visits = 16384000ull;  // the C3 hop-2 request
  const uint32_t vpw = 10 * 4;          // visits per wave (4 segments of 10)
  const uint64_t waves = visits / vpw;
  char* table; f4* out;
  // NB up to 2^16 rows per block: round the table up to whole blocks
  const uint64_t alloc_rows = (table_rows + 65535) & ~65535ull;
  CHECK(hipMalloc(&table, alloc_rows * 1024));
  CHECK(hipMemset(table, 1, alloc_rows * 1024));
  CHECK(hipMalloc(&out, waves * 64 * sizeof(f4)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("# %llu random 1 KiB rows of a %.1f GB table per launch; median of 5 launches; TB/s = gathered bytes / time\n",
         (unsigned long long)visits, table_rows * 1024 / 1e9);
  printf("%6s %8s %12s %9s %7s\n", "piece", "NB", "stride", "ms", "TB/s");
  const int pieces[] = {10, 9, 8, 7};  // 1024 (contiguous), 512, 256, 128 bytes
  for (int pl : pieces) {
    for (int nbl = 0; nbl <= 16; ++nbl) {
      if (pl == 10 && nbl > 0) break;
      // whole blocks only: rows beyond the last whole block are not visited
      const uint64_t rows = (table_rows >> nbl) << nbl;
      std::vector<float> ms;
      for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        gather<10><<<(unsigned)(waves / 4), 256>>>(table, rows, (uint32_t)pl, (uint32_t)nbl, vpw, 0x1234 + rep * 7919ull, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t; CHECK(hipEventElapsedTime(&t, e0, e1));
        if (rep) ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      const double m = ms[ms.size() / 2];
      printf("%6d %8d %12llu %9.3f %7.2f\n", 1 << pl, 1 << nbl, (unsigned long long)1 << (pl + nbl), m,
             visits * 1024.0 / (m * 1e-3) / 1e12);
      fflush(stdout);
    }
  }
  return 0;
}
