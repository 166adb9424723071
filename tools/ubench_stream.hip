// Streaming-read ceiling on MI355X for the access shapes of the cull pass: N x 16-byte loads per lane
// from one or two arrays, nothing written.  Prints achieved TB/s for several sizes and load depths.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int kPerLane, bool kTwo, bool kNT>
__global__ __launch_bounds__(256) void rd(const v4f* __restrict__ a, const v4f* __restrict__ b, unsigned n, float* sink) {
  const unsigned base = blockIdx.x * 256u * kPerLane + threadIdx.x;
  v4f x[kPerLane], y[kPerLane];
#pragma unroll
  for (int k = 0; k < kPerLane; ++k) {
    const unsigned i = base + k * 256u;
    const unsigned ic = i < n ? i : 0u;
    x[k] = kNT ? __builtin_nontemporal_load(a + ic) : a[ic];
    if (kTwo) y[k] = kNT ? __builtin_nontemporal_load(b + ic) : b[ic];
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < kPerLane; ++k) s += x[k].x + x[k].w + (kTwo ? y[k].y + y[k].z : 0.0f);
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int kPerLane, bool kTwo, bool kNT>
static void run(const char* name, const v4f* a, const v4f* b, unsigned n, float* sink) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const unsigned blocks = (n + 256u * kPerLane - 1) / (256u * kPerLane);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((rd<kPerLane, kTwo, kNT>), dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rd<kPerLane, kTwo, kNT>), dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)n * 16 * (kTwo ? 2 : 1);
  printf("%-40s n=%9u  %7.1f us  %5.2f TB/s\n", name, n, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
  const unsigned nmax = 1u << 25;
  v4f *a, *b; float* sink;
  (void)hipMalloc(&a, (size_t)nmax * 16 * 3); (void)hipMalloc(&b, (size_t)nmax * 16 * 3); (void)hipMalloc(&sink, 4096);
  (void)hipMemset(a, 0, (size_t)nmax * 16 * 3); (void)hipMemset(b, 0, (size_t)nmax * 16 * 3);
  for (unsigned n : {1u << 22, 1u << 24}) {
    run<4, true, true>("2 arrays, 4 loads each, nontemporal", a, b, n, sink);
    run<4, true, false>("2 arrays, 4 loads each", a, b, n, sink);
    run<8, true, true>("2 arrays, 8 loads each, nontemporal", a, b, n, sink);
    run<2, true, true>("2 arrays, 2 loads each, nontemporal", a, b, n, sink);
    run<8, false, true>("1 array (2x length), 8 loads, nontemporal", a, b, 2 * n, sink);
    run<4, false, true>("1 array (2x length), 4 loads, nontemporal", a, b, 2 * n, sink);
  }
  return 0;
}
