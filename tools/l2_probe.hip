// l2_probe.hip - does a layer's weight slab, touched by the launch BEFORE, arrive sooner? (round 6)
//
// A small pass of b18c384nbt is ~122 dependent launches; its 3x3 layers read 663 552 bytes of weights each, 48 MB per pass - more than the
// 32 MB of L2 the chip has, and every XCD's work-groups read their own copy: each layer's weights arrive from the Infinity Cache / HBM, and
// the small shapes' chunks are as long as that round trip. This probe measures what the NEXT layer's weights cost when the
// CURRENT launch has already pulled them into its XCD's L2:
//   chains of `layers` dependent launches, each of `wgs` work-groups x 512 threads with the small shapes' access pattern (waves 0-3 each read
//   the work-group's 110 592-byte slice - a cout group of 32 - in 6 chunks of 18 lines of 16 bytes per lane, a chunk requested one chunk
//   ahead of its use);
//   mode 0: nothing else; mode 1: waves 4-7 touch one dword per 128-byte line of the NEXT layer's weights - all the lines the work-groups on
//   this XCD will read (work-group b of the next launch runs on XCD b % 8: observed, MI355X_MICROARCH.md) - split over the XCD's work-groups;
//   mode 2: the same lines, but touched from the WRONG XCD (rotated by one): what the placement is worth; mode 3: every layer is launched
//   TWICE (the second launch of a layer finds what the first one read, if anything survives a kernel boundary).
// Output per mode: us per launch (hipEvents over the chain), and from s_memtime stamps (100 MHz) of wave 0 of every work-group: the time to
// the first chunk's arrival and the time of the six chunks.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/l2_probe tools/l2_probe.hip ; run: tools/_build/l2_probe [wgs ...]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int LAYER_BYTES = 192 * 192 * 9 * 2;  // 663 552
constexpr int SLICE_BYTES = LAYER_BYTES / 6;     // a cout group of 32: 110 592
constexpr int CHUNKS = 6;
constexpr int LINES_PER_CHUNK = SLICE_BYTES / CHUNKS / (64 * 16);  // 18 x 16 bytes per lane

__device__ __forceinline__ unsigned xccId() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

template <int MODE>
__global__ __launch_bounds__(512) void layerKernel(const char* w, const char* wNext, unsigned* sink, unsigned long long* stamps, int slot) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned acc = 0;
  if(wave < 4) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    const u32x4* src = (const u32x4*)(w + (size_t)(blockIdx.x % 6) * SLICE_BYTES) + lane;
    u32x4 buf[2][LINES_PER_CHUNK];
#pragma unroll
    for(int i = 0; i < LINES_PER_CHUNK; i++) buf[0][i] = src[(size_t)i * 64];
    unsigned long long t1 = 0;
#pragma unroll
    for(int c = 0; c < CHUNKS; c++) {
      if(c + 1 < CHUNKS) {
#pragma unroll
        for(int i = 0; i < LINES_PER_CHUNK; i++)
          buf[(c + 1) & 1][i] = src[(size_t)((c + 1) * LINES_PER_CHUNK + i) * 64];
      }
#pragma unroll
      for(int i = 0; i < LINES_PER_CHUNK; i++) acc ^= buf[c & 1][i][0] ^ buf[c & 1][i][3];
      if(c == 0) {
        asm volatile("" ::"v"(acc));
        t1 = __builtin_readcyclecounter();
      }
    }
    asm volatile("" ::"v"(acc));
    const unsigned long long t2 = __builtin_readcyclecounter();
    if(threadIdx.x == 0) {
      stamps[((size_t)slot * gridDim.x + blockIdx.x) * 2 + 0] = t1 - t0;
      stamps[((size_t)slot * gridDim.x + blockIdx.x) * 2 + 1] = t2 - t0;
    }
  }
  else if((MODE == 1 || MODE == 2) && wNext != nullptr) {
    // the lines of the next layer that THIS XCD's work-groups will read, split over this XCD's work-groups
    const unsigned me = MODE == 1 ? blockIdx.x % 8 : (blockIdx.x + 1) % 8;  // whose lines (mode 2: the neighbour's)
    const int rank = blockIdx.x / 8, peers = (gridDim.x - (blockIdx.x % 8) + 7) / 8;
    // work-groups of XCD `me` in the next launch: b = me, me + 8, ...; their slices are (b % 6)
    unsigned need = 0;
    for(int b = me; b < (int)gridDim.x; b += 8) need |= 1u << (b % 6);
    const int t = threadIdx.x - 256;
    const int ns = __builtin_popcount(need);
    // k-th needed slice -> slice number, packed 3 bits each
    unsigned packed = 0;
    for(int s = 5; s >= 0; s--)
      if(need >> s & 1u) packed = packed << 3 | (unsigned)s;
    constexpr int LPS = SLICE_BYTES / 128;  // lines per slice: 864
    const int total = ns * LPS;
    // all of a thread's requests in flight before the first is consumed (batches of 8)
    for(int i0 = rank * 256 + t; i0 < total; i0 += peers * 256 * 8) {
      unsigned v[8];
#pragma unroll
      for(int u = 0; u < 8; u++) {
        const int i = i0 + u * peers * 256;
        v[u] = 0;
        if(i < total) v[u] = *(const volatile unsigned*)(wNext + (size_t)(packed >> (3 * (i / LPS)) & 7u) * SLICE_BYTES + (size_t)(i % LPS) * 128);
      }
#pragma unroll
      for(int u = 0; u < 8; u++) acc ^= v[u];
    }
  }
  if(acc == 0x12345678u) sink[0] = acc;  // keeps the loads
}

static double runMode(int mode, int wgs, int layers, int iters, const char* w, unsigned* sink, unsigned long long* stamps, double* firstUs, double* loopUs) {
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto chain = [&] {
    int slot = 0;
    for(int l = 0; l < layers; l++) {
      const char* wl = w + (size_t)l * LAYER_BYTES;
      const char* wn = l + 1 < layers ? wl + LAYER_BYTES : nullptr;
      const int reps = mode == 3 ? 2 : 1;
      for(int r = 0; r < reps; r++) {
        if(mode == 1) hipLaunchKernelGGL(layerKernel<1>, dim3(wgs), dim3(512), 0, st, wl, wn, sink, stamps, slot);
        else if(mode == 2) hipLaunchKernelGGL(layerKernel<2>, dim3(wgs), dim3(512), 0, st, wl, wn, sink, stamps, slot);
        else hipLaunchKernelGGL(layerKernel<0>, dim3(wgs), dim3(512), 0, st, wl, wn, sink, stamps, slot);
        slot++;
      }
    }
  };
  for(int i = 0; i < 3; i++) chain();
  CHECK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, st));
  for(int i = 0; i < iters; i++) chain();
  CHECK(hipEventRecord(e1, st));
  CHECK(hipStreamSynchronize(st));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int slots = layers * (mode == 3 ? 2 : 1);
  std::vector<unsigned long long> h((size_t)slots * wgs * 2);
  CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  // mode 3: report the SECOND launch of every layer; skip the first layer of a chain (nobody prefetched it)
  double f = 0, t = 0;
  long n = 0;
  for(int s = (mode == 3 ? 3 : 1); s < slots; s += (mode == 3 ? 2 : 1))
    for(int b = 0; b < wgs; b++) {
      f += (double)h[((size_t)s * wgs + b) * 2];
      t += (double)h[((size_t)s * wgs + b) * 2 + 1];
      n++;
    }
  *firstUs = f / n / 100.0;  // s_memtime: 100 MHz
  *loopUs = t / n / 100.0;
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  CHECK(hipStreamDestroy(st));
  return (double)ms * 1e3 / ((double)iters * slots);
}

int main(int argc, char** argv) {
  const int layers = 72, iters = 20;
  char* w = nullptr;
  unsigned* sink = nullptr;
  unsigned long long* stamps = nullptr;
  CHECK(hipMalloc(&w, (size_t)layers * LAYER_BYTES));
  CHECK(hipMemset(w, 1, (size_t)layers * LAYER_BYTES));
  CHECK(hipMalloc(&sink, 64));
  std::vector<int> wgsList;
  for(int i = 1; i < argc; i++) wgsList.push_back(atoi(argv[i]));
  if(wgsList.empty()) wgsList = {18, 48, 192};
  for(int wgs : wgsList) {
    CHECK(hipMalloc(&stamps, (size_t)layers * 2 * wgs * 2 * 8));
    const char* names[4] = {"no prefetch", "next layer touched from its own XCD", "next layer touched from the WRONG XCD", "every layer launched twice (2nd launch)"};
    for(int rep = 0; rep < 2; rep++)
      for(int mode = 0; mode < 4; mode++) {
        double f = 0, t = 0;
        const double us = runMode(mode, wgs, layers, iters, w, sink, stamps, &f, &t);
        printf("[l2 probe] %3d work-groups, %-42s: %6.2f us per launch; first chunk after %5.2f us, six chunks after %5.2f us\n", wgs, names[mode], us, f, t);
        fflush(stdout);
      }
    CHECK(hipFree(stamps));
  }
  return 0;
}
