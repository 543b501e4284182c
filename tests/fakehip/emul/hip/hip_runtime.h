// TEST INFRASTRUCTURE: a host-side stand-in for <hip/hip_runtime.h> under which the plain-C++ device kernels of
// katago_amd/csrc/transformer_kernels.hip compile for x86 and RUN on the CPU (tests/fakehip/emulate_transformer.cpp):
// a work-group is a set of OS threads, __syncthreads a barrier (emu::Barrier), __shfl_xor an exchange through a per-wave array,
// dynamic LDS a static buffer, work-groups run one after the other. Only what those kernels use is provided.
#pragma once
#define KMX_EMULATED_HIP 1  // engine.cpp: no virtual-memory API here (its KMX_DEBUG_GUARD placement is a device-only triage mode)
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorInvalidDevice = 101, hipErrorNotReady = 600 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
// the host runtime, for the whole-engine emulation (emulate_engine.cpp): device memory is host memory, streams are no-ops
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };
inline const char* hipGetErrorString(hipError_t) { return "emulated HIP"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceGetPCIBusId(char* id, int len, int) {  // no PCI device behind the emulation: numa.cpp then leaves placement alone
  if(len > 0) id[0] = 0;
  return hipErrorInvalidValue;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "CPU emulation");
  strcpy(p->gcnArchName, "none");
  return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
// launches execute immediately here, so there is nothing to capture: refusing makes the engine launch directly
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return 801; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 801; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 801; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 801; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = malloc(8); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = malloc(8); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = malloc(8); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

namespace emu {
struct Idx { unsigned x, y, z; };
// A barrier whose participants may leave (a thread that returns from the kernel). One 64-bit word holds (expected, arrived), so
// exactly one arrival or departure completes a phase; waiters sleep on the phase word through the futex system call directly - a mutex +
// condition variable makes every one of hundreds of woken lanes queue for the mutex again (two thirds of the CPU suite's time was system
// time), and std::barrier of libstdc++ waits through a shared pool of atomics, which costs a lot with hundreds of threads and makes
// ThreadSanitizer see synchronisation between unrelated barriers. ThreadSanitizer sees this one through its acquire / release atomics.
class Barrier {
 public:
  explicit Barrier(unsigned n) : state_((uint64_t)n << 32) {}
  void reset(unsigned n) { state_.store((uint64_t)n << 32, std::memory_order_release); }  // (only while nobody uses it)
  void arrive_and_wait() {
    const uint32_t ph = phase_.load(std::memory_order_acquire);
    uint64_t s = state_.load(std::memory_order_relaxed);
    for(;;) {
      const uint32_t expected = (uint32_t)(s >> 32), arrived = (uint32_t)s + 1;
      const bool last = arrived == expected;
      if(state_.compare_exchange_weak(s, last ? (uint64_t)expected << 32 : s + 1, std::memory_order_acq_rel, std::memory_order_relaxed)) {
        if(last) release();
        else
          while(phase_.load(std::memory_order_acquire) == ph) syscall(SYS_futex, (uint32_t*)&phase_, FUTEX_WAIT_PRIVATE, ph, nullptr, nullptr, 0);
        return;
      }
    }
  }
  void arrive_and_drop() {
    uint64_t s = state_.load(std::memory_order_relaxed);
    for(;;) {
      const uint32_t expected = (uint32_t)(s >> 32) - 1, arrived = (uint32_t)s;
      const bool last = expected > 0 && arrived == expected;
      if(state_.compare_exchange_weak(s, (uint64_t)expected << 32 | (last ? 0u : arrived), std::memory_order_acq_rel, std::memory_order_relaxed)) {
        if(last) release();
        return;
      }
    }
  }

 private:
  void release() {  // (nobody can arrive for the next phase before the phase word moves: everyone else is asleep on it or gone)
    phase_.fetch_add(1, std::memory_order_release);
    syscall(SYS_futex, (uint32_t*)&phase_, FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
  }
  std::atomic<uint64_t> state_;  // expected << 32 | arrived
  std::atomic<uint32_t> phase_{0};
};
// The exchange arrays of a wave's collectives (MFMA operands, shuffles) exist twice and a lane counts its collectives (waveOp): collective
// k writes set k % 2, meets the wave's barrier ONCE and reads. A lane can only write set k % 2 again (collective k + 2) after the barrier of
// collective k + 1, at which every lane has arrived with its reads of collective k behind it - so no second barrier per collective.
struct Wave {
  float buf[2][64];
  float opA[2][64][8], opB[2][64][8];
  std::unique_ptr<Barrier> bar;
};
struct Block {
  std::unique_ptr<Barrier> bar;
  std::vector<Wave> waves;
};
extern thread_local Idx tIdx, bIdx, bDim, gDim;
extern thread_local Block* cur;
extern thread_local bool dropped;
extern thread_local unsigned waveOp;  // collectives this lane has taken part in (all lanes of a wave execute the same sequence)
void* dynLds();
void launchImpl(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::tIdx)
#define blockIdx (emu::bIdx)
#define blockDim (emu::bDim)
#define gridDim (emu::gDim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* const var = (type*)emu::dynLds();
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu::launchImpl(grid, block, [&] { kern(__VA_ARGS__); })
// v_mfma_f32_32x32x16_{f16,bf16} as a wave-collective on the CPU: D[32x32] = A[32x16] B[16x32] + C with the register
// layout of the hardware (the layout the GPU-verified convolution kernel relies on): lane l holds A[l%32][8*(l/32)..+7],
// B[8*(l/32)..+7][l%32], and C/D element v of lane l is row (v/4)*8 + (l/32)*4 + v%4, column l%32.
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma16(a, b, c)
// wave / work-group builtins of the convolution kernel (emulate_engine.cpp, "real convolution" build)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_barrier() __syncthreads()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

namespace emu {
template <class V8, class F16>
inline F16 mfma16(V8 a, V8 b, F16 c) {
  Wave& w = cur->waves[tIdx.x >> 6];
  const unsigned lane = tIdx.x & 63, set = waveOp++ & 1;
  float(*const opA)[8] = w.opA[set];
  float(*const opB)[8] = w.opB[set];
  for(int i = 0; i < 8; i++) {
    opA[lane][i] = (float)a[i];
    opB[lane][i] = (float)b[i];
  }
  w.bar->arrive_and_wait();
  const unsigned col = lane & 31, half = lane >> 5;
  for(int v = 0; v < 16; v++) {
    const unsigned row = (v / 4) * 8 + half * 4 + (v % 4);
    float sum = 0.0f;
    for(int k = 0; k < 16; k++) sum += opA[row + 32 * (k / 8)][k % 8] * opB[col + 32 * (k / 8)][k % 8];
    c[v] += sum;
  }
  return c;
}
}  // namespace emu

namespace emu {
inline void vmAtBarrier();
int lateDmaMode();
}
inline void __syncthreads() {
  if(emu::lateDmaMode() == 2) {
    // every lane has arrived - whatever the phase before the barrier reads has been read - THEN the required copies land, and only
    // then does anyone go on
    emu::cur->bar->arrive_and_wait();
    emu::vmAtBarrier();
  }
  emu::cur->bar->arrive_and_wait();
}
namespace emu {
// Lanes of a wave execute in lockstep on the hardware, so a wave may write LDS and read other lanes' values back without a
// barrier; OS threads do not. Kernels that rely on it get this call injected at those points (see the convolution's
// epilogue in tests/test_engine_emulated.py).
inline void waveSync() { cur->waves[tIdx.x >> 6].bar->arrive_and_wait(); }
// global_load_lds: every lane copies `size` bytes from its own global address to (wave-uniform LDS base) + lane * size.
// On the hardware the copy completes asynchronously: a wave's vector-memory requests retire IN ORDER, and s_waitcnt vmcnt(N) returns
// when at most N of them are outstanding. The emulation runs the two legal extremes:
//   * immediate (default): the copy happens at issue - the earliest the hardware could complete it (a slot that is overwritten while
//     another wave still reads it shows up, thread interleaving permitting; tools/emulated_tsan.sh);
//   * KMX_EMU_LATE_DMA=1: the copy happens only when an s_waitcnt of the issuing wave forces it - the LATEST the hardware may complete
//     it. A count that is too generous (vmcnt(N) with N one too large, a ring one slot too shallow for the requests in flight, a
//     barrier that publishes a slab its requester has not waited for) leaves the destination stale and the results wrong.
// Requests that are not LDS-DMA (plain global loads and stores count in vmcnt on gfx9 too) are entered with vmNote() where a
// kernel's counts rely on them.
struct VmOp { const void* src; void* dst; int size; bool reg = false; };  // reg: a plain load into the lane's own registers
struct VmQueue {
  std::vector<VmOp> ops;
  size_t head = 0;     // ops[head ..) are outstanding, oldest first
  size_t required = 0;  // mode 2: ops[.. required) must have landed - a wait has said so - and do at the lane's next barrier
};
extern thread_local VmQueue vmQueue;  // per lane (a lane is an OS thread here; all lanes of a wave issue and wait alike)
// KMX_EMU_LATE_DMA: 0 - a copy lands when it is issued; 1 - when an s_waitcnt vmcnt(N) of its wave forces it (the latest the HARDWARE may
// complete it); 2 - at the barrier that FOLLOWS that wait (or at the wave's exit): the latest moment ANOTHER wave may legally first see it -
// what wait + barrier publish. Lanes are OS threads here and a fetching wave reaches its next wait long before a multiplying wave has
// read anything, so mode 1 can hide a copy that is required one step too late; mode 2 cannot. (Mode 2 is only for kernels in which no
// wave reads what it fetched itself before a barrier.)
int lateDmaMode();
inline bool lateDma() { return lateDmaMode() != 0; }
inline void vmLandUpTo(size_t end) {  // ops[head .. end) land
  VmQueue& q = vmQueue;
  while(q.head < end) {
    const VmOp& op = q.ops[q.head++];
    if(op.dst != nullptr) memcpy(op.dst, op.src, (size_t)op.size);
  }
  if(q.head == q.ops.size()) {
    q.ops.clear();
    q.head = 0;
    q.required = 0;
  }
}
inline void waitVm(int n) {
  const int mode = lateDmaMode();
  if(mode == 0) return;
  VmQueue& q = vmQueue;
  const size_t end = q.ops.size() > (size_t)n ? q.ops.size() - (size_t)n : 0;
  if(mode == 1) vmLandUpTo(end > q.head ? end : q.head);
  else {
    // mode 2 delays what OTHER waves may see; a load into the lane's own registers (globalLoadReg) is there for its own wave once the wait
    // returns
    for(size_t i = q.head; i < end; i++)
      if(q.ops[i].reg && q.ops[i].dst != nullptr) {
        memcpy(q.ops[i].dst, q.ops[i].src, (size_t)q.ops[i].size);
        q.ops[i].dst = nullptr;
      }
    if(end > q.required) q.required = end;
  }
}
inline void vmAtBarrier() {  // mode 2: what the waits so far have required lands now
  if(lateDmaMode() == 2 && vmQueue.required > vmQueue.head) vmLandUpTo(vmQueue.required);
}
inline void vmNote() {  // a vector-memory request that is not an LDS-DMA copy: it only takes its place in the in-order queue
  if(lateDma()) vmQueue.ops.push_back(VmOp{nullptr, nullptr, 0});
}
// a plain global load into registers whose s_waitcnt the KERNEL writes (conv_small_kernel.h REGW: gloadFrag / waitFrag): under the late
// modes the destination keeps its old contents until a wait of the lane forces the load - a count one too generous multiplies a stale
// fragment
inline void globalLoadReg(void* dst, const void* src, int size) {
  if(lateDma()) {
    VmOp op{src, dst, size};
    op.reg = true;
    vmQueue.ops.push_back(op);
  }
  else memcpy(dst, src, (size_t)size);
}
template <class G, class L>
inline void globalLoadLds(G gsrc, L ldsBase, int size, int, int) {
  void* const dst = (char*)(uintptr_t)ldsBase + (tIdx.x & 63) * size;
  if(lateDma()) vmQueue.ops.push_back(VmOp{(const void*)(uintptr_t)gsrc, dst, size});
  else memcpy(dst, (const void*)(uintptr_t)gsrc, (size_t)size);
}
}  // namespace emu
// v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second; returns {new first, new second}
namespace emu {
struct Pair32 {
  unsigned v[2];
  unsigned operator[](int i) const { return v[i]; }
  template <class V> operator V() const { V r; r[0] = v[0]; r[1] = v[1]; return r; }
};
inline Pair32 permlane32Swap(unsigned a, unsigned b) {
  Wave& w = cur->waves[tIdx.x >> 6];
  const unsigned lane = tIdx.x & 63, set = waveOp++ & 1;
  unsigned ua, ub;
  memcpy(&w.opA[set][lane][0], &a, 4);
  memcpy(&w.opB[set][lane][0], &b, 4);
  w.bar->arrive_and_wait();
  Pair32 r;
  if(lane < 32) {
    r.v[0] = a;
    memcpy(&ua, &w.opA[set][lane + 32][0], 4);
    r.v[1] = ua;  // new second, lower half = old first, upper half
  }
  else {
    memcpy(&ub, &w.opB[set][lane - 32][0], 4);
    r.v[0] = ub;  // new first, upper half = old second, lower half
    r.v[1] = b;
  }
  return r;
}
}  // namespace emu
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emu::permlane32Swap(a, b)
inline float __shfl_xor(float v, int laneMask) {
  emu::Wave& w = emu::cur->waves[emu::tIdx.x >> 6];
  const unsigned lane = emu::tIdx.x & 63, set = emu::waveOp++ & 1;
  w.buf[set][lane] = v;
  w.bar->arrive_and_wait();
  return w.buf[set][lane ^ (unsigned)laneMask];
}
