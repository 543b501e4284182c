// numa.h — host placement of a device's helper threads and pinned buffers ("NUMA placement of server threads and pinned
// buffers" is what limits the 1 -> 8 GPU curve of a path that has no collective).
//
// The reference runs one NNEvaluator server thread per GPU in one process (cpp/neuralnet/nneval.cpp:399-407, device choice
// cpp/program/setup.cpp:174-220) and leaves placement to the OS. Here each device's leaf batcher has two helper threads (dispatcher,
// completion) and pinned staging; on a two-socket node the wrong socket costs every staged row and every result a trip over the
// inter-socket link. So, per device:
//   * its NUMA node is read from sysfs (/sys/bus/pci/devices/<pci bus id>/numa_node, hipDeviceGetPCIBusId);
//   * the batcher's dispatcher and completion threads restrict their affinity to that node's CPUs (intersected with what the process
//     may use: a container's cpuset stays in force);
//   * pinned buffers: hipHostMalloc places host memory on the node closest to the CURRENT device unless hipHostMallocNumaUser is given,
//     and every Engine allocates with its own device current - so the runtime already does the right thing; the allocating thread
//     additionally prefers the device's node (set_mempolicy MPOL_PREFERRED around the allocations), which is what places the pages if a
//     runtime build does not. nodeOfAddress() lets a test read back where a buffer landed.
// KATAMX_NUMA = on (default) | off; KATAMX_NUMA_VERBOSE=1 prints one line per binding to stderr (the 8-fake-device dry run reads
// them); KMX_SYSFS_ROOT replaces "/sys" (tests build a fake two-node tree there). Nothing here fails a run: an unknown topology
// (numa_node = -1, no sysfs, a single node) leaves everything as the OS placed it.
#ifndef KMX_NUMA_H_
#define KMX_NUMA_H_

#include <string>
#include <vector>

namespace kmx {
namespace numa {

bool enabled();
// NUMA node of a HIP device, or -1 when unknown (no PCI id, no sysfs entry, the kernel reports -1)
int nodeOfDevice(int device);
// CPUs of a node (sysfs cpulist), empty when unknown
std::vector<int> cpusOfNode(int node);
// "0-3,8,10-11" -> {0,1,2,3,8,10,11}
std::vector<int> parseCpuList(const std::string& s);
// Restrict the calling thread to the CPUs of `node` that it may already run on. Returns the number of CPUs it is bound to, 0 when
// nothing was changed (unknown node, empty intersection, NUMA handling off). `what` names the thread in the verbose line.
int bindThisThreadToNode(int node, const char* what, int device);
// Memory policy of the calling thread: prefer `node` for pages it faults in from now on (node < 0: back to the default policy).
// Returns false when the kernel refused or nothing was done.
bool preferNodeForThisThread(int node);
// the node a resident page lies on (move_pages query), or -1
int nodeOfAddress(const void* p);

// scope guard: prefer the device's node for the allocations inside the scope
class PreferDeviceNode {
 public:
  explicit PreferDeviceNode(int device);
  ~PreferDeviceNode();
  int node() const { return node_; }

 private:
  int node_ = -1;
  bool set_ = false;
};

}  // namespace numa
}  // namespace kmx
#endif
