// numa.cpp — see numa.h. Linux only (sysfs, sched_setaffinity, the mempolicy system calls by number: libnuma is not a dependency).
#include "numa.h"

#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace kmx {
namespace numa {

namespace {
std::string sysfsRoot() {
  const char* e = getenv("KMX_SYSFS_ROOT");
  return e != nullptr && e[0] != 0 ? std::string(e) : std::string("/sys");
}
bool verbose() {
  const char* e = getenv("KATAMX_NUMA_VERBOSE");
  return e != nullptr && e[0] == '1';
}
bool readLine(const std::string& path, std::string* out) {
  std::ifstream f(path);
  if(!f) return false;
  std::getline(f, *out);
  return true;
}
std::string formatCpus(const std::vector<int>& cpus) {
  std::ostringstream o;
  for(size_t i = 0; i < cpus.size();) {
    size_t j = i;
    while(j + 1 < cpus.size() && cpus[j + 1] == cpus[j] + 1) j++;
    if(i) o << ",";
    o << cpus[i];
    if(j > i) o << "-" << cpus[j];
    i = j + 1;
  }
  return o.str();
}
// MPOL_* of <linux/mempolicy.h>
constexpr int kMpolDefault = 0, kMpolPreferred = 1;
}  // namespace

bool enabled() {
  const char* e = getenv("KATAMX_NUMA");
  return !(e != nullptr && (strcmp(e, "off") == 0 || strcmp(e, "0") == 0));
}

std::vector<int> parseCpuList(const std::string& s) {
  std::vector<int> out;
  size_t i = 0;
  while(i < s.size()) {
    while(i < s.size() && !isdigit((unsigned char)s[i])) i++;
    if(i >= s.size()) break;
    int a = 0;
    while(i < s.size() && isdigit((unsigned char)s[i])) a = a * 10 + (s[i++] - '0');
    int b = a;
    if(i < s.size() && s[i] == '-') {
      i++;
      b = 0;
      while(i < s.size() && isdigit((unsigned char)s[i])) b = b * 10 + (s[i++] - '0');
    }
    for(int c = a; c <= b && c < 65536; c++) out.push_back(c);
  }
  std::sort(out.begin(), out.end());
  out.erase(std::unique(out.begin(), out.end()), out.end());
  return out;
}

int nodeOfDevice(int device) {
  if(device < 0) return -1;
  char bus[64] = {0};
  if(hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess || bus[0] == 0) return -1;
  std::string id(bus);
  for(char& c : id) c = (char)tolower((unsigned char)c);  // sysfs spells the hex digits in lower case
  std::string line;
  if(!readLine(sysfsRoot() + "/bus/pci/devices/" + id + "/numa_node", &line)) return -1;
  const int node = atoi(line.c_str());
  return node < 0 ? -1 : node;
}

std::vector<int> cpusOfNode(int node) {
  if(node < 0) return {};
  std::string line;
  if(!readLine(sysfsRoot() + "/devices/system/node/node" + std::to_string(node) + "/cpulist", &line)) return {};
  return parseCpuList(line);
}

int bindThisThreadToNode(int node, const char* what, int device) {
  if(!enabled() || node < 0) return 0;
  const std::vector<int> cpus = cpusOfNode(node);
  if(cpus.empty()) return 0;
  cpu_set_t have, want;
  CPU_ZERO(&have);
  if(sched_getaffinity(0, sizeof(have), &have) != 0) return 0;
  CPU_ZERO(&want);
  std::vector<int> bound;
  for(int c : cpus)
    if(c < CPU_SETSIZE && CPU_ISSET(c, &have)) {
      CPU_SET(c, &want);
      bound.push_back(c);
    }
  if(bound.empty()) return 0;  // the process may not run on that node at all (a cpuset elsewhere): leave the thread where it is
  if(sched_setaffinity(0, sizeof(want), &want) != 0) return 0;
  if(verbose()) fprintf(stderr, "[katamx numa] device %d -> node %d: %s thread bound to cpus %s\n", device, node, what, formatCpus(bound).c_str());
  return (int)bound.size();
}

bool preferNodeForThisThread(int node) {
  if(!enabled()) return false;
#ifdef SYS_set_mempolicy
  if(node < 0) return syscall(SYS_set_mempolicy, kMpolDefault, nullptr, 0UL) == 0;
  if(node >= 1024) return false;
  unsigned long mask[1024 / (8 * sizeof(unsigned long))];
  memset(mask, 0, sizeof(mask));
  mask[node / (8 * sizeof(unsigned long))] |= 1UL << (node % (8 * sizeof(unsigned long)));
  return syscall(SYS_set_mempolicy, kMpolPreferred, mask, (unsigned long)(sizeof(mask) * 8)) == 0;
#else
  (void)node;
  return false;
#endif
}

int nodeOfAddress(const void* p) {
#ifdef SYS_move_pages
  void* pages[1] = {(void*)((uintptr_t)p & ~(uintptr_t)4095)};
  int status[1] = {-1};
  if(syscall(SYS_move_pages, 0, 1UL, pages, nullptr, status, 0) != 0) return -1;
  return status[0] < 0 ? -1 : status[0];
#else
  (void)p;
  return -1;
#endif
}

PreferDeviceNode::PreferDeviceNode(int device) {
  if(!enabled()) return;
  node_ = nodeOfDevice(device);
  if(node_ < 0) return;
#ifdef SYS_get_mempolicy
  // a policy the user gave the process (numactl --interleave ...) is not ours to replace: act only on the default policy
  int mode = kMpolDefault;
  if(syscall(SYS_get_mempolicy, &mode, nullptr, 0UL, nullptr, 0UL) != 0 || mode != kMpolDefault) return;
#endif
  set_ = preferNodeForThisThread(node_);
  if(verbose())
    fprintf(stderr, "[katamx numa] device %d -> node %d: pinned staging %s\n", device, node_,
            set_ ? "allocated with that node preferred" : "left to the runtime's placement (the kernel refused the node)");
}
PreferDeviceNode::~PreferDeviceNode() {
  if(set_) preferNodeForThisThread(-1);
}

}  // namespace numa
}  // namespace kmx
