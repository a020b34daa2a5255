// grid_sync_chain.hip — what would ONE launch per alignment cost against a launch per evaluation?  The skeleton of a lone pair's
// fine-level evaluation (pixel loop + block fold 3.6 us, record fold over G records, solve 2.4 us) in two forms:
//   chain       N launches; every block first folds the previous launch's G records and "solves" (k_iterate's structure), then
//               "evaluates" and writes its record
//   persistent  ONE launch of G co-resident blocks; per evaluation: "evaluate", record written with returning agent-scope
//               exchanges, a ticket; the block that draws the last ticket folds the G records (agent-scope loads), "solves",
//               publishes the state and bumps an epoch; the others poll the epoch (bounded spins)
// The arithmetic is replaced by waits of the measured durations (s_memrealtime, 100 MHz), so that only the synchronisation
// differs.  hipcc --offload-arch=gfx950 tools/ubench/grid_sync_chain.hip -o tools/ubench/grid_sync_chain && ./grid_sync_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kBlock = 256, kRecWords64 = 32;

__device__ __forceinline__ void busy_wait_ticks(unsigned ticks) {   // 100 MHz ticks
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((unsigned)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
}

// fold of G records by a whole block: thread (slot = tid & 31, part = tid >> 5) adds slot `slot` of records part, part + 8, ...
template <bool COHERENT>
__device__ __forceinline__ double fold_records(const unsigned long long* recs, int G, double* lds) {
  const int tid = threadIdx.x, slot = tid & 31, part = tid >> 5;
  double s = 0.0;
  for (int q = part; q < G; q += 8) {
    unsigned long long v;
    if (COHERENT) v = __hip_atomic_load(recs + (size_t)q * kRecWords64 + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else v = recs[(size_t)q * kRecWords64 + slot];
    s += __longlong_as_double((long long)v);
  }
  lds[tid] = s;
  __syncthreads();
  double t = 0.0;
  if (tid < 32)
    for (int p = 0; p < 8; p++) t += lds[p * 32 + tid];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(kBlock) void k_chain(unsigned long long* recs_in, unsigned long long* recs_out, int G, unsigned work_ticks,
                                                  unsigned solve_ticks, double* sink) {
  __shared__ double lds[kBlock];
  const double t = fold_records<false>(recs_in, G, lds);
  if (threadIdx.x < 64) busy_wait_ticks(solve_ticks);     // the solve: one wave's dependent chain, repeated by every block
  __syncthreads();
  busy_wait_ticks(work_ticks);                            // pixel loop + block fold
  if (threadIdx.x < 32) recs_out[(size_t)blockIdx.x * kRecWords64 + threadIdx.x] = (unsigned long long)__double_as_longlong(t * 1e-9 + (double)blockIdx.x);
  if (blockIdx.x == 0 && threadIdx.x == 0) *sink = t;
}

struct Sync { unsigned int ticket; unsigned int epoch; unsigned int failed; unsigned int pad; double state[4]; };

__global__ __launch_bounds__(kBlock) void k_persistent(unsigned long long* recs, Sync* sy, int G, int N, unsigned work_ticks, unsigned solve_ticks,
                                                       double* sink) {
  __shared__ double lds[kBlock];
  __shared__ int s_last;
  double state = 0.0;
  for (int it = 0; it < N; it++) {
    busy_wait_ticks(work_ticks);                          // pixel loop + block fold at the current state
    if (threadIdx.x < 32) {                               // the record, to the coherence point, and this wave knows it is there
      const unsigned long long old = __hip_atomic_exchange(recs + (size_t)blockIdx.x * kRecWords64 + threadIdx.x,
                                                           (unsigned long long)__double_as_longlong(state * 1e-9 + (double)blockIdx.x),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" ::"v"(old) : "memory");
    }
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&sy->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u;
    __syncthreads();
    if (s_last) {                                         // block-uniform: this block folds, solves, publishes
      if (threadIdx.x == 0) __hip_atomic_store(&sy->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double t = fold_records<true>(recs, G, lds);
      if (threadIdx.x < 64) busy_wait_ticks(solve_ticks);
      if (threadIdx.x < 4) {
        const unsigned long long old = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(&sy->state[threadIdx.x]),
                                                             (unsigned long long)__double_as_longlong(t + it), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old) : "memory");
      }
      if (threadIdx.x == 0) __hip_atomic_store(&sy->epoch, (unsigned)it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // everybody (the publisher too: its own store is what it finds) waits for the epoch; bounded
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(&sy->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it + 1u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000u) { __hip_atomic_store(&sy->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
    state = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&sy->state[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (__hip_atomic_load(&sy->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *sink = state;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int N = 40;
  unsigned long long *ra, *rb; Sync* sy; double* sink;
  CHK(hipMalloc(&ra, 256 * 256)); CHK(hipMalloc(&rb, 256 * 256)); CHK(hipMalloc(&sy, sizeof(Sync))); CHK(hipMalloc(&sink, 8));
  CHK(hipMemset(ra, 0, 256 * 256)); CHK(hipMemset(rb, 0, 256 * 256));
  hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  std::printf("skeleton of %d evaluations: work 3.6 us, solve 2.4 us (waits), records of 256 B; us per evaluation\n", N);
  std::printf("%6s %14s %14s\n", "blocks", "chain", "persistent");
  for (int G : {16, 32, 64, 150, 256}) {
    float best_c = 1e9f, best_p = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      CHK(hipEventRecord(e0, s));
      for (int it = 0; it < N; it++)
        hipLaunchKernelGGL(k_chain, dim3(G), dim3(kBlock), 0, s, (it & 1) ? rb : ra, (it & 1) ? ra : rb, G, 360u, 240u, sink);
      CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_c) best_c = ms;
      CHK(hipMemsetAsync(sy, 0, sizeof(Sync), s));
      CHK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_persistent, dim3(G), dim3(kBlock), 0, s, ra, sy, G, N, 360u, 240u, sink);
      CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
      CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best_p) best_p = ms;
      Sync h; CHK(hipMemcpy(&h, sy, sizeof(h), hipMemcpyDeviceToHost));
      if (h.failed || h.epoch != (unsigned)N) { std::printf("persistent form failed: epoch %u failed %u\n", h.epoch, h.failed); return 2; }
    }
    std::printf("%6d %14.2f %14.2f\n", G, best_c * 1e3f / N, best_p * 1e3f / N);
  }
  return 0;
}
