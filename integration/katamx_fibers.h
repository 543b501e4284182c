// katamx_fibers.h — K leaves in flight per OS thread for the reference's UNMODIFIED search.
//
// The reference's search descends the tree recursively and calls NNEvaluator::evaluate from the bottom of that recursion
// (cpp/search/search.cpp:1189-1463 -> searchnnhelpers.cpp:61-135); evaluate returns when the result is there, so a leaf in
// flight costs one blocked OS thread, and filling a 256-row device batch takes 256+ kernel threads that wake and sleep once
// per leaf. Virtual losses already make concurrent descents of one tree meaningful (search.cpp:1349,1396); what is missing
// is a way to have several descents per OS thread.
//
// Here a "search thread" of the reference (one invocation of the task that Search::performTaskWithThreads fans out,
// cpp/search/searchmultithreadhelpers.cpp:77-92, with its own SearchThread state, search.cpp:567-620) runs on a FIBER: a
// user-level context with its own stack (2 MB of address space, committed on touch). K fibers share an OS thread. When a fiber's leaf has been handed to the device
// (KatamxNNEval::begin returned with the leaf in flight), this repo's NNEvaluator::evaluate parks the fiber instead of
// blocking; the OS thread continues with its next fiber - which descends again, sees the virtual losses of the parked
// descents and submits another leaf - and only when all of its fibers are parked does it block, on the OLDEST ticket.
// numSearchThreads = 512 then means 512 descents in flight on 64 OS threads (KATAMX_LEAVES_PER_THREAD=8): the search
// semantics are exactly those of the reference with 512 search threads; only the carrier of a descent changes.
//
// Linking: oracle/Makefile links katago_hipx / katago_oraclex with
//     -Wl,--wrap=_ZN6Search22performTaskWithThreadsEPSt8functionIFviEEi
// so that the calls of search.cpp reach performTaskOnFibers below; nothing under /root/reference is edited. With
// KATAMX_LEAVES_PER_THREAD unset or 1 the wrapper forwards to the reference's own thread pool.
#ifndef KATAMX_FIBERS_H_
#define KATAMX_FIBERS_H_

#include <cstdint>
#include <functional>

namespace KatamxLeaf { struct Port; }

namespace KatamxFibers {

// KATAMX_LEAVES_PER_THREAD (1 ... 64; default 1 = fibers off), read once.
int leavesPerThread();
// Is the calling code running on a fiber of this module?
bool onFiber();
// For a leaf that is on the device. On a fiber: parks it; this OS thread runs its other fibers and, when none can run, blocks
// on the oldest parked ticket. Returns true once THIS ticket has been waited for (KatamxLeaf::wait was called for it here -
// each ticket exactly once - and rethrows what that wait threw). Not on a fiber: returns false at once.
bool park(KatamxLeaf::Port* port, uint64_t ticket);
// Lets every other fiber of this OS thread that can make progress run once (parked ones are waited for, in order).
// Returns false when there is none or the caller is not on a fiber.
bool yieldToOthers();
// Runs (*task)(indices[0]) ... (*task)(indices[count-1]) as fibers on the calling OS thread; returns when all have returned.
// The first exception a task threw is rethrown after the others have finished.
void runOnFibers(std::function<void(int)>* task, const int* indices, int count);
// counters since process start (tests / logs): fibers run, parks, blocking waits of an OS thread
void counters(uint64_t& fibersRun, uint64_t& parks, uint64_t& blockingWaits);

}  // namespace KatamxFibers

#endif
