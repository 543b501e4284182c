// clock_meter.hip - the shader clock of the chip WHILE something else runs on it (round 6).
//
// One wave on its own stream samples s_memtime (shader-clock ticks) against s_memrealtime (100 MHz) every `period` microseconds for `seconds`
// seconds and sleeps in between (s_sleep): it occupies one wave slot of one CU. Run it beside a workload of ANOTHER process - `bench.py`,
// `katago_hip selfplay` - and it says what clock the chip sustained under that workload: the figure the MFMA peak has to be scaled by before
// a kernel's `roofline.frac` says anything about the kernel (the nominal 2.5 PFLOP/s are 2.4 GHz).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/clock_meter tools/clock_meter.hip
//   run:   tools/_build/clock_meter <seconds> <period_us> [csv]      prints mean / min / max and deciles of the per-interval clock
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ __launch_bounds__(64) void meterKernel(unsigned long long* out, int samples, unsigned long long periodTicks) {
  if(threadIdx.x != 0) return;
  for(int i = 0; i < samples; i++) {
    const unsigned long long r0 = wall_clock64();
    out[2 * i] = __builtin_readcyclecounter();
    out[2 * i + 1] = r0;
    while(wall_clock64() - r0 < periodTicks) __builtin_amdgcn_s_sleep(32);
  }
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
  const double periodUs = argc > 2 ? atof(argv[2]) : 200.0;
  const char* csv = argc > 3 ? argv[3] : nullptr;
  const int samples = (int)(seconds * 1e6 / periodUs);
  unsigned long long* d = nullptr;
  CHECK(hipMalloc(&d, (size_t)samples * 16));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipLaunchKernelGGL(meterKernel, dim3(1), dim3(64), 0, st, d, samples, (unsigned long long)(periodUs * 100.0));
  CHECK(hipStreamSynchronize(st));
  std::vector<unsigned long long> h((size_t)samples * 2);
  CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> mhz;
  FILE* f = csv ? fopen(csv, "w") : nullptr;
  if(f) fprintf(f, "ms,shader_mhz\n");
  for(int i = 1; i < samples; i++) {
    const double dr = (double)(h[2 * i + 1] - h[2 * i - 1]), dc = (double)(h[2 * i] - h[2 * i - 2]);
    if(dr <= 0) continue;
    mhz.push_back(dc / dr * 100.0);
    if(f) fprintf(f, "%.3f,%.0f\n", (double)(h[2 * i + 1] - h[1]) / 1e5, mhz.back());
  }
  if(f) fclose(f);
  if(mhz.empty()) return 1;
  std::vector<double> s = mhz;
  std::sort(s.begin(), s.end());
  double sum = 0;
  for(double v : s) sum += v;
  printf("[clock meter] %zu intervals of %.0f us: shader clock mean %.0f MHz, min %.0f, max %.0f; deciles", s.size(), periodUs, sum / s.size(), s.front(), s.back());
  for(int q = 1; q < 10; q++) printf(" %.0f", s[s.size() * q / 10]);
  printf("\n");
  // mean per second of the run (a workload that starts and stops beside the meter shows as a step)
  const int perSec = (int)(1e6 / periodUs);
  printf("[clock meter] mean per second:");
  for(size_t i = 0; i + perSec <= mhz.size(); i += perSec) {
    double a = 0;
    for(int k = 0; k < perSec; k++) a += mhz[i + k];
    printf(" %.0f", a / perSec);
  }
  printf("\n");
  return 0;
}
