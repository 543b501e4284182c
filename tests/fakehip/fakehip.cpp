// fakehip.cpp — TEST INFRASTRUCTURE: a stand-in for the HIP runtime that lets libkatamx.so build and "run" its launch
// schedule on a machine without a GPU. Loaded with LD_PRELOAD ahead of libamdhip64 by tests/test_schedule_dryrun.py.
//   * device memory is host memory (hipMalloc = calloc, hipMemcpy = memcpy), streams and events are no-ops;
//   * kernels are never executed: every launch is LOGGED (kernel name as registered by the fat binary, grid, block,
//     dynamic LDS bytes and the bytes of the argument struct) to the file named by KMX_FAKEHIP_LOG;
//   * pointers inside argument structs are rewritten as  <buffer ordinal by first appearance in the log>+<offset>, and
//     the first time a buffer appears its size and an FNV-1a hash of its contents are logged too — so two builds that
//     issue the same launches on the same data produce the same text, wherever the allocator put things.
// Nothing in the product links or loads this file.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../katago_amd/csrc/kernels.h"  // argument struct sizes only (pulls in <hip/hip_runtime.h> for dim3, hipError_t)

namespace {
std::mutex g_mu;
struct Alloc { size_t size; int ordinal; int dev; };
// (constructed on first use, never destroyed: when this library is preloaded into a BINARY that links libkatamx.so - the dry run of
// katago_hip on fake devices - the fat-binary registration of libkatamx.so's static initialisers calls in here before this
// library's own static initialisers have run)
std::map<uintptr_t, Alloc>& allocsRef() { static auto* m = new std::map<uintptr_t, Alloc>(); return *m; }  // base -> info (device allocations only)
std::map<const void*, std::string>& namesRef() { static auto* m = new std::map<const void*, std::string>(); return *m; }  // host stub -> kernel name
#define g_allocs allocsRef()
#define g_names namesRef()
int g_nextOrdinal = 0;
FILE* g_log = nullptr;
// Several fake devices (KMX_FAKEHIP_DEVICES=N, default 1): every allocation, stream and event belongs to the device that was
// current when it was created, and every later use is checked against the calling thread's current device - the rule the real
// runtime enforces for launches, event records and stream waits. A mismatch is logged as a VIOLATION line; with one device
// nothing below changes the log (tests/golden/schedule_md5.json).
int numDevices() {
  static const int n = [] {
    const char* e = getenv("KMX_FAKEHIP_DEVICES");
    const int v = e ? atoi(e) : 1;
    return v < 1 ? 1 : v;
  }();
  return n;
}
thread_local int t_dev = 0;
std::map<const void*, int>& ownerRef() { static auto* m = new std::map<const void*, int>(); return *m; }  // stream / event -> device
#define g_owner ownerRef()
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<CallCfg> g_cfg;

FILE* logFile() {
  if(!g_log) {
    const char* p = getenv("KMX_FAKEHIP_LOG");
    g_log = p ? fopen(p, "w") : stderr;
    if(!g_log) g_log = stderr;
  }
  return g_log;
}
void checkOwner(const char* what, const void* res) {
  if(numDevices() == 1 || res == nullptr) return;
  std::lock_guard<std::mutex> l(g_mu);
  auto it = g_owner.find(res);
  if(it != g_owner.end() && it->second != t_dev) {
    fprintf(logFile(), "VIOLATION %s: resource of device %d used while device %d is current\n", what, it->second, t_dev);
    fflush(logFile());
  }
}
void own(const void* res) {
  std::lock_guard<std::mutex> l(g_mu);
  g_owner[res] = t_dev;
}
uint64_t fnv(const void* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  const unsigned char* b = (const unsigned char*)p;
  for(size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
// demangled-ish short name: keep the mangled string (stable across builds of the same source)
size_t argBytes(const std::string& name) {
  using namespace kmx;
  if(name.find("convMfmaKernel") != std::string::npos || name.find("convSmallKernel") != std::string::npos) return sizeof(ConvArgs);
  if(name.find("convChainKernel") != std::string::npos) return sizeof(ConvChainArgs);
  if(name.find("inputExpandKernel") != std::string::npos) return sizeof(InputArgs);
  if(name.find("gpoolApply") != std::string::npos) return sizeof(GPoolArgs);
  if(name.find("policyFinal") != std::string::npos) return sizeof(PolicyArgs);
  if(name.find("valueFinal") != std::string::npos) return sizeof(ValueArgs);
  if(name.find("bnActKernel") != std::string::npos) return sizeof(BnActArgs);
#ifdef KMX_FAKEHIP_TRANSFORMER
  if(name.find("rmsNormKernel") != std::string::npos) return sizeof(RmsNormArgs);
  if(name.find("attentionKernel") != std::string::npos || name.find("attentionMfmaKernel") != std::string::npos) return sizeof(AttentionArgs);
  if(name.find("swiGluKernel") != std::string::npos) return sizeof(SwiGluArgs);
#endif
  return 0;  // several scalar arguments: not dumped
}
// pointer -> "B<ordinal>+<offset>" if it points into a device allocation
bool describePtr(uint64_t v, char* out, size_t outLen, FILE* f) {
  if(v == 0) return false;
  auto it = g_allocs.upper_bound((uintptr_t)v);
  if(it == g_allocs.begin()) return false;
  --it;
  if(v >= it->first + it->second.size) return false;
  if(it->second.ordinal < 0) {
    it->second.ordinal = g_nextOrdinal++;
    fprintf(f, "  buffer B%d size %zu hash %016llx\n", it->second.ordinal, it->second.size,
            (unsigned long long)fnv((const void*)it->first, it->second.size));
  }
  snprintf(out, outLen, "B%d+%llu", it->second.ordinal, (unsigned long long)(v - it->first));
  return true;
}
}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = numDevices(); return hipSuccess; }
hipError_t hipSetDevice(int d) {
  if(d < 0 || d >= numDevices()) return hipErrorInvalidDevice;
  t_dev = d;
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) { *d = t_dev; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* prop, int) {  // what hipGetDeviceProperties is a macro for
  memset(prop, 0, sizeof(*prop));
  prop->multiProcessorCount = 256;
  return hipSuccess;
}
// PCI bus id of fake device d: bus 0x1A + d, upper-case hex as the real runtime prints it (numa.cpp lower-cases it for sysfs); the NUMA
// tests build a matching tree under KMX_SYSFS_ROOT (tests/test_schedule_dryrun.py)
hipError_t hipDeviceGetPCIBusId(char* pciBusId, int len, int device) {
  if(device < 0 || device >= numDevices() || len < 13) return hipErrorInvalidValue;
  snprintf(pciBusId, (size_t)len, "0000:%02X:00.0", 0x1A + device);
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t) { return "fakehip"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) {
  *p = calloc(1, n ? n : 1);
  std::lock_guard<std::mutex> l(g_mu);
  g_allocs[(uintptr_t)*p] = Alloc{n, -1, t_dev};
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if(p) {
    std::lock_guard<std::mutex> l(g_mu);
    g_allocs.erase((uintptr_t)p);
    free(p);
  }
  return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { checkOwner("hipMemcpyAsync", st); memcpy(d, s, n); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = calloc(1, n ? n : 1); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = (hipStream_t)calloc(1, 8); own(*s); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)calloc(1, 8); own(*s); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { checkOwner("hipStreamSynchronize", s); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { checkOwner("hipStreamQuery", s); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t, unsigned int) { checkOwner("hipStreamWaitEvent", s); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned int) { *e = (hipEvent_t)calloc(1, 8); own(*e); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)calloc(1, 8); own(*e); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free((void*)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { checkOwner("hipEventRecord (event)", e); checkOwner("hipEventRecord (stream)", s); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
// the dry run logs launches as they are issued: capture is refused, the engine then launches directly
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

// ---- fat binary registration: only the host stub -> kernel name map is kept ----
void** __hipRegisterFatBinary(const void*) { static void* dummy[4]; return dummy; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void* hostFunction, char*, const char* deviceName, unsigned int, void*, void*, void*, void*, int*) {
  std::lock_guard<std::mutex> l(g_mu);
  g_names[hostFunction] = deviceName ? deviceName : "?";
}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void*, void**, void*, const char*, size_t, unsigned) {}
void __hipRegisterSurface(void**, void*, char*, char*, int, int) {}
void __hipRegisterTexture(void**, void*, char*, char*, int, int, int) {}

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
  g_cfg.push_back(CallCfg{grid, block, shmem, stream});
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
  CallCfg c = g_cfg.back();
  g_cfg.pop_back();
  *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
  return hipSuccess;
}
// KMX_FAKEHIP_QUIET=1: launches are only COUNTED per device (one summary line per device at exit) - the host-capacity measurement
// (tests/test_host_capacity.py) runs the whole host stack against a device that costs nothing, and formatting and hashing every
// launch under one lock would be what it measures.
static bool quietMode() {
  static const bool q = [] { const char* e = getenv("KMX_FAKEHIP_QUIET"); return e != nullptr && e[0] == '1'; }();
  return q;
}
static std::atomic<unsigned long long> g_quietLaunches[64];
static void quietSummary() {
  for(int d = 0; d < numDevices() && d < 64; d++) fprintf(logFile(), "dev %d launches %llu\n", d, g_quietLaunches[d].load());
  fflush(logFile());
}
hipError_t hipLaunchKernel(const void* func, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t stream) {
  checkOwner("hipLaunchKernel", stream);
  if(quietMode()) {
    static const int registered = (atexit(quietSummary), 0);
    (void)registered;
    g_quietLaunches[t_dev < 64 ? t_dev : 63].fetch_add(1, std::memory_order_relaxed);
    return hipSuccess;
  }
  std::lock_guard<std::mutex> l(g_mu);
  FILE* f = logFile();
  auto it = g_names.find(func);
  const std::string name = it == g_names.end() ? "?" : it->second;
  const size_t nb = argBytes(name);
  std::string line;
  if(nb > 0 && args && args[0]) {
    const unsigned char* a = (const unsigned char*)args[0];
    char tmp[64];
    for(size_t off = 0; off < nb; off += 8) {
      const size_t w = nb - off >= 8 ? 8 : nb - off;
      uint64_t v = 0;
      memcpy(&v, a + off, w);
      if(w == 8 && describePtr(v, tmp, sizeof(tmp), f)) line += std::string(" ") + tmp;
      else { snprintf(tmp, sizeof(tmp), " %llx", (unsigned long long)v); line += tmp; }
    }
  }
  if(numDevices() > 1) fprintf(f, "dev %d ", t_dev);
  fprintf(f, "launch %s grid %u,%u,%u block %u lds %zu args%s\n", name.c_str(), grid.x, grid.y, grid.z, block.x, shmem, line.c_str());
  fflush(f);
  return hipSuccess;
}

}  // extern "C"
