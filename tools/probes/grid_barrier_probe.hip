// grid_barrier_probe.hip — what does ONE device-wide barrier cost on MI355X, in the form VERDICT r5 item 4 asked to be measured
// for a one-kernel conv + BN-statistics + apply?  Each workgroup publishes 2*co partial sums with agent-scope
// (sc1) stores, crosses a grid barrier built from a per-XCD arrival counter + one cross-XCD hop (no __threadfence, s_sleep polling),
// reads back the totals' rows, and goes on.  Variants: flat (one counter for all workgroups) | two-level (per-XCD counters, the last
// arriver of an XCD bumps the global one).  Every poll loop is BOUNDED (a workgroup that never sees the release sets an error flag
// and leaves): the probe cannot hang the GPU.  The grid must be co-resident (<= one workgroup per CU here).
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Ctl {
  unsigned int xcd_count[8][32];     // one 128-byte line per XCD
  unsigned int global_count[32];
  unsigned int release[32];          // generation number
  unsigned int error[32];
};

__device__ __forceinline__ unsigned int ld_agent(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// returns false when the bounded poll ran out
template <bool TWO_LEVEL>
__device__ __forceinline__ bool grid_barrier(Ctl* c, unsigned int gen, int n_wg, int wg_per_xcd) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (TWO_LEVEL) {
      const int x = blockIdx.x & 7;                       // workgroups are dealt round-robin to the 8 XCDs
      const unsigned int a = __hip_atomic_fetch_add(&c->xcd_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a + 1 == gen * (unsigned)wg_per_xcd) {
        const unsigned int g = __hip_atomic_fetch_add(&c->global_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g + 1 == gen * 8u) __hip_atomic_store(&c->release[0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const unsigned int g = __hip_atomic_fetch_add(&c->global_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g + 1 == gen * (unsigned)n_wg) __hip_atomic_store(&c->release[0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int polls = 0;
    while (ld_agent(&c->release[0]) < gen) {
      __builtin_amdgcn_s_sleep(2);
      if (++polls > (1 << 20)) { ok = false; __hip_atomic_store(&c->error[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
  return ok;
}

// iters x [publish 2*co floats (sc1) -> barrier -> read all n_wg rows of this thread's column back]
template <bool TWO_LEVEL>
__global__ __launch_bounds__(256) void probe_kernel(Ctl* c, float* rows, int co2, int iters, unsigned long long* t_out, float* sink, int readback) {
  const int n_wg = gridDim.x, wg_per_xcd = n_wg / 8;
  const unsigned long long t0 = wall_clock64();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // two partial-row slots alternate so that a fast workgroup's next publish cannot overwrite a row a slow one still reads
    float* slot = rows + (size_t)(it & 1) * n_wg * co2;
    for (int i = threadIdx.x; i < co2; i += 256)
      __hip_atomic_store(slot + (size_t)blockIdx.x * co2 + i, (float)(it + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_barrier<TWO_LEVEL>(c, (unsigned)(it + 1), n_wg, wg_per_xcd)) break;
    if (readback) {
      // every workgroup re-reduces the rows it needs (like imm_bn_apply_fused: fixed order, the same totals everywhere)
      for (int i = threadIdx.x; i < co2; i += 256) {
        float s = 0.f;
        for (int w = 0; w < n_wg; ++w) s += __hip_atomic_load(slot + (size_t)w * co2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += s;
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("# %s, %d CUs; wall_clock64 at 100 MHz\n", p.gcnArchName, p.multiProcessorCount);
  Ctl* ctl; float* rows; unsigned long long* t; float* sink;
  const int max_wg = 256, max_co2 = 1024;
  CHECK(hipMalloc(&ctl, sizeof(Ctl)));
  CHECK(hipMalloc(&rows, (size_t)2 * max_wg * max_co2 * 4));
  CHECK(hipMalloc(&t, max_wg * 8));
  CHECK(hipMalloc(&sink, 4));
  printf("# variant     workgroups  2*co  readback   us per [publish + barrier (+ read back)]   (median over workgroups of total / iters)\n");
  const int iters = 200;
  for (int two = 0; two < 2; ++two)
    for (int n_wg : {64, 128, 256})
      for (int co2 : {0, 256, 512})
        for (int rb = 0; rb < 2; ++rb) {
          if (co2 == 0 && rb) continue;
          CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
          for (int rep = 0; rep < 2; ++rep) {               // second repetition is the measurement (first: code load)
            CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
            if (two) hipLaunchKernelGGL(probe_kernel<true>, dim3(n_wg), dim3(256), 0, 0, ctl, rows, co2, iters, t, sink, rb);
            else hipLaunchKernelGGL(probe_kernel<false>, dim3(n_wg), dim3(256), 0, 0, ctl, rows, co2, iters, t, sink, rb);
            CHECK(hipDeviceSynchronize());
          }
          Ctl h;
          CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
          std::vector<unsigned long long> ht(n_wg);
          CHECK(hipMemcpy(ht.data(), t, n_wg * 8, hipMemcpyDeviceToHost));
          std::sort(ht.begin(), ht.end());
          printf("%-11s %10d %5d %9s   %8.2f%s\n", two ? "two-level" : "flat", n_wg, co2, rb ? "yes" : "no",
                 (double)ht[n_wg / 2] / 100.0 / iters, h.error[0] ? "   (POLL LIMIT HIT: grid not co-resident?)" : "");
        }
  // for scale: an empty kernel launched back to back on one stream (the kernel boundary the barrier would replace)
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe_kernel<false>, dim3(256), dim3(256), 0, 0, ctl, rows, 0, 0, t, sink, 0);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("# 200 empty 256-workgroup launches back to back (eager, one stream): %.2f us per launch\n", ms * 1e3 / 200);
  return 0;
}
