// glx memory-system probes: the achievable ceilings the roofline lines of bench.py are priced against
// ("measure an achievable peak with a device-to-device copy / stream-triad kernel").
// Not part of the sampling / aggregation path: hand-written streaming and gather kernels whose byte
// counts are known exactly, timed with HIP events on the caller's stream.
//   STREAM_READ  every lane reads 16 B per step, grid-strided, reduces in registers    bytes moved = n
//   COPY         b[i] = a[i]                                                            2n
//   TRIAD        a[i] = b[i] + s * c[i]   (McCalpin's STREAM triad)                      3n
//   GATHER32     one aligned 32-byte record per draw from uniformly random positions of a table
//                (the access EdgeWeightSampler makes per output slot: alias_method.cc:117-121 on one
//                packed record), 16 B written per draw (nbr + eid), coalesced      records/s, 48 B each
//   GATHER_ROWS  whole feature rows (row_bytes each) from uniformly random rows, one group of lanes per
//                row, 16 B per lane, reduced in registers (the access of the segmented reduce:
//                aggregator.cc:25-59 without its arithmetic)                        rows * row_bytes
#include "glx_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void glx_probe_read_kernel(const f4* __restrict__ a, int64_t n4, float* sink) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  // four independent 16-byte loads in flight per lane
  for (; i + 3 * step < n4; i += 4 * step) {
    const f4 v0 = __builtin_nontemporal_load(a + i);
    const f4 v1 = __builtin_nontemporal_load(a + i + step);
    const f4 v2 = __builtin_nontemporal_load(a + i + 2 * step);
    const f4 v3 = __builtin_nontemporal_load(a + i + 3 * step);
    acc += (v0 + v1) + (v2 + v3);
  }
  for (; i < n4; i += step) acc += __builtin_nontemporal_load(a + i);
  const float s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s == 123.456f) sink[0] = s;  // never true for the zero-filled buffer: keeps the loads alive
}

__global__ __launch_bounds__(256) void glx_probe_copy_kernel(const f4* __restrict__ a, f4* __restrict__ b, int64_t n4) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i + 3 * step < n4; i += 4 * step) {  // four loads in flight before the first store
    const f4 v0 = __builtin_nontemporal_load(a + i);
    const f4 v1 = __builtin_nontemporal_load(a + i + step);
    const f4 v2 = __builtin_nontemporal_load(a + i + 2 * step);
    const f4 v3 = __builtin_nontemporal_load(a + i + 3 * step);
    __builtin_nontemporal_store(v0, b + i);
    __builtin_nontemporal_store(v1, b + i + step);
    __builtin_nontemporal_store(v2, b + i + 2 * step);
    __builtin_nontemporal_store(v3, b + i + 3 * step);
  }
  for (; i < n4; i += step) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}

__global__ __launch_bounds__(256) void glx_probe_triad_kernel(f4* __restrict__ a, const f4* __restrict__ b,
                                                              const f4* __restrict__ c, float s, int64_t n4) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i + step < n4; i += 2 * step) {
    const f4 x0 = __builtin_nontemporal_load(b + i);
    const f4 y0 = __builtin_nontemporal_load(c + i);
    const f4 x1 = __builtin_nontemporal_load(b + i + step);
    const f4 y1 = __builtin_nontemporal_load(c + i + step);
    __builtin_nontemporal_store(x0 + s * y0, a + i);
    __builtin_nontemporal_store(x1 + s * y1, a + i + step);
  }
  for (; i < n4; i += step) {
    const f4 x = __builtin_nontemporal_load(b + i);
    const f4 y = __builtin_nontemporal_load(c + i);
    __builtin_nontemporal_store(x + s * y, a + i);
  }
}

struct Rec32 {
  int64_t a, b, c, d;
};

// One thread = two draws (like glx_sample_slots_kernel: one Philox block per slot pair), 32 B out per thread.
__global__ __launch_bounds__(256) void glx_probe_gather32_kernel(const Rec32* __restrict__ table, uint64_t num_records,
                                                                 int64_t pairs, uint64_t salt, longlong2* __restrict__ out) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= pairs) return;
  const uint64_t h0 = glx_mix64((uint64_t)t * 2 + salt);
  const uint64_t h1 = glx_mix64((uint64_t)t * 2 + 1 + salt);
  const uint64_t i0 = (uint64_t)(((unsigned __int128)h0 * num_records) >> 64);
  const uint64_t i1 = (uint64_t)(((unsigned __int128)h1 * num_records) >> 64);
  const Rec32 r0 = table[i0];
  const Rec32 r1 = table[i1];
  out[2 * t] = longlong2{r0.a + r0.c, r1.a + r1.c};
  out[2 * t + 1] = longlong2{r0.b + r0.d, r1.b + r1.d};
}

// G lanes per row, U rows in flight per lane group: the gather of the segmented reduce without the reduce.
template <int G, int U>
__global__ __launch_bounds__(256) void glx_probe_gather_rows_kernel(const f4* __restrict__ table, uint64_t num_rows,
                                                                    int32_t row_f4, int64_t groups, int32_t rows_per_group,
                                                                    uint64_t salt, f4* __restrict__ out) {
  const int64_t gid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int c = threadIdx.x & (G - 1);
  if (gid >= groups) return;
  for (int32_t col = c; col < row_f4; col += G) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int32_t base = 0; base < rows_per_group; base += U) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t h = glx_mix64((uint64_t)(gid * rows_per_group + base + u) + salt);
        const uint64_t r = (uint64_t)(((unsigned __int128)h * num_rows) >> 64);
        v[u] = (base + u < rows_per_group) ? table[r * row_f4 + col] : f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
    out[gid * row_f4 + col] = acc;
  }
}

}  // namespace

extern "C" int glx_probe_bandwidth(int device, int kind, int64_t bytes, int64_t units, int32_t unit_bytes, int32_t reps,
                                   double* moved_bytes_out, double* avg_ms_out, void* stream) {
  GLX_REQUIRE(moved_bytes_out && avg_ms_out, "NULL output");
  GLX_REQUIRE(kind >= GLX_PROBE_STREAM_READ && kind <= GLX_PROBE_GATHER_ROWS, "unknown probe kind %d", kind);
  GLX_REQUIRE(bytes >= (1 << 20) && bytes % 4096 == 0, "bytes must be a multiple of 4096 and >= 1 MiB");
  GLX_REQUIRE(reps >= 1 && reps <= 1000, "reps must be in [1, 1000]");
  int rc = glx_init_device(device);
  if (rc != GLX_OK) return rc;
  GlxDeviceGuard guard(device);
  GLX_REQUIRE(guard.ok, "cannot select device %d", device);
  hipStream_t s = glx_stream(stream);
  const int64_t n4 = bytes / 16;
  GlxTemp a, b, c;
  GLX_HIP(hipMalloc(&a.p, (size_t)bytes));
  GLX_HIP(hipMemsetAsync(a.p, 0, (size_t)bytes, s));
  double moved = 0.0;
  int64_t out_bytes = 0;
  int32_t row_f4 = 0, G = 64, rows_per_group = 10;
  int64_t groups = 0;
  if (kind == GLX_PROBE_COPY || kind == GLX_PROBE_TRIAD) {
    GLX_HIP(hipMalloc(&b.p, (size_t)bytes));
    GLX_HIP(hipMemsetAsync(b.p, 0, (size_t)bytes, s));
  }
  if (kind == GLX_PROBE_TRIAD) {
    GLX_HIP(hipMalloc(&c.p, (size_t)bytes));
    GLX_HIP(hipMemsetAsync(c.p, 0, (size_t)bytes, s));
  }
  if (kind == GLX_PROBE_STREAM_READ) {
    GLX_HIP(hipMalloc(&b.p, 256));
    moved = (double)bytes;
  } else if (kind == GLX_PROBE_COPY) {
    moved = 2.0 * (double)bytes;
  } else if (kind == GLX_PROBE_TRIAD) {
    moved = 3.0 * (double)bytes;
  } else if (kind == GLX_PROBE_GATHER32) {
    GLX_REQUIRE(units >= 2 && units % 2 == 0, "GATHER32: units = number of draws (even)");
    out_bytes = units * 16;
    GLX_HIP(hipMalloc(&b.p, (size_t)out_bytes));
    moved = (double)units * 48.0;
  } else {
    GLX_REQUIRE(unit_bytes >= 16 && unit_bytes % 16 == 0 && bytes % unit_bytes == 0,
                "GATHER_ROWS: unit_bytes = row bytes (multiple of 16 dividing bytes)");
    GLX_REQUIRE(units >= rows_per_group, "GATHER_ROWS: units = rows gathered per launch");
    row_f4 = unit_bytes / 16;
    G = row_f4 >= 64 ? 64 : (row_f4 >= 32 ? 32 : 16);
    groups = units / rows_per_group;
    out_bytes = groups * (int64_t)unit_bytes;
    GLX_HIP(hipMalloc(&b.p, (size_t)out_bytes));
    moved = (double)groups * rows_per_group * unit_bytes + (double)out_bytes;
  }
  hipEvent_t e0, e1;
  GLX_HIP(hipEventCreate(&e0));
  GLX_HIP(hipEventCreate(&e1));
  // enough workgroups to fill 256 CUs x 8 waves several times over; grid-strided
  const unsigned stream_grid = 256 * 16;
  for (int rep = -2; rep < reps; ++rep) {
    if (rep == 0) (void)hipEventRecord(e0, s);
    const uint64_t salt = 0x9E3779B97F4A7C15ull * (uint64_t)(rep + 3);
    switch (kind) {
      case GLX_PROBE_STREAM_READ:
        glx_probe_read_kernel<<<stream_grid, 256, 0, s>>>(a.as<f4>(), n4, b.as<float>());
        break;
      case GLX_PROBE_COPY:
        glx_probe_copy_kernel<<<stream_grid, 256, 0, s>>>(a.as<f4>(), b.as<f4>(), n4);
        break;
      case GLX_PROBE_TRIAD:
        glx_probe_triad_kernel<<<stream_grid, 256, 0, s>>>(a.as<f4>(), b.as<f4>(), c.as<f4>(), 3.0f, n4);
        break;
      case GLX_PROBE_GATHER32: {
        const int64_t pairs = units / 2;
        glx_probe_gather32_kernel<<<(unsigned)((pairs + 255) / 256), 256, 0, s>>>(a.as<Rec32>(), (uint64_t)(bytes / 32), pairs,
                                                                                 salt, b.as<longlong2>());
        break;
      }
      default: {
        const int64_t threads = groups * G;
        const unsigned grid = (unsigned)((threads + 255) / 256);
        const uint64_t nrows = (uint64_t)(bytes / unit_bytes);
        if (G == 64) glx_probe_gather_rows_kernel<64, 10><<<grid, 256, 0, s>>>(a.as<f4>(), nrows, row_f4, groups, rows_per_group, salt, b.as<f4>());
        else if (G == 32) glx_probe_gather_rows_kernel<32, 10><<<grid, 256, 0, s>>>(a.as<f4>(), nrows, row_f4, groups, rows_per_group, salt, b.as<f4>());
        else glx_probe_gather_rows_kernel<16, 10><<<grid, 256, 0, s>>>(a.as<f4>(), nrows, row_f4, groups, rows_per_group, salt, b.as<f4>());
        break;
      }
    }
  }
  (void)hipEventRecord(e1, s);
  hipError_t le = hipGetLastError();
  hipError_t se = hipEventSynchronize(e1);
  float ms = 0.f;
  if (le == hipSuccess && se == hipSuccess) se = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  GLX_HIP(le);
  GLX_HIP(se);
  *moved_bytes_out = moved;
  *avg_ms_out = (double)ms / reps;
  return GLX_OK;
}
