// batcher.cpp — the persistent leaf batcher behind the C ABI (include/katamx.h, kmx_batcher_*).
//
// Replaces, for callers that adopt it, the server half of the reference's NNEvaluator (cpp/neuralnet/nneval.cpp:562-752:
// serve() popping up to maxBatch NNResultBuf* from a ThreadsafeQueue, cpp/core/threadsafequeue.h:173-189, and calling
// NeuralNet::getOutput synchronously) and the row copies of the backend's getOutput:
//   * kmx_batcher_submit is called by the SEARCH thread that owns the leaf. It bit-packs the row's feature planes (1012 instead
//     of 31768 bytes per 19x19 row; the reference's binaryInputNCHWPacked layout; kmx_batcher_submit_packed takes
//     them already packed from a caller that featurises into bits), reserves a row in the batch that is filling and copies
//     the bits into that batch's PINNED staging - in parallel with the other submitters, outside the lock.
//   * a dispatcher thread seals the filling batch when the device is idle (greedy like waitPopUpToN: it never waits for more
//     rows once one is waiting and the device has nothing to do) or when it is full (then up to `max_in_flight` batches are
//     between H2D and D2H, each on its own engine and stream, so that H2D of batch k+1, the kernels of batch k and D2H of
//     batch k-1 overlap); rows that arrive while the device is busy accumulate into the next batch instead of queueing
//     behind a synchronous call.
//   * a completion thread waits for each batch's event, copies every row's results out of pinned memory into the buffers its
//     submitter named, frees the staging set and wakes exactly the threads whose rows were in it. A staging set is therefore
//     never held by uncollected results: a thread with several tickets outstanding (a server thread that submitted a whole
//     batch, a search thread with several leaves) can always submit more - submit() only ever waits for the DEVICE.
//   * kmx_batcher_wait blocks until the ticket's row is done.
// Rows/batches counters have the meaning of nneval.cpp:712-713. Rows never interact, so a row's outputs are bit-identical to
// the same row through kmx_eval, whatever batch it lands in (tests/test_gpu_batcher.py).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/katamx.h"
#include "engine.h"
#include "model_desc.h"
#include "numa.h"

namespace kmx {

class Batcher {
 public:
  Batcher(const ModelDesc& model, int nnXLen, int nnYLen, int maxBatch, int dtype, int device, int maxInFlight, int numSlots)
    : maxBatch_(maxBatch), maxInFlight_(maxInFlight) {
    S_ = nnXLen * nnYLen;
    // Seal at the device's granule. The convolutions give a board to a work-group and a work-group to a CU: a pass over 430 boards
    // on 256 CUs costs what a pass over 512 does. A filling batch therefore counts as FULL at ONE granule - the device's CU count
    // (max_batch_size itself below that) - and goes behind the running one at once, instead of growing to an odd size while the device
    // is busy: the reference's benchmark at 1024 leaves in flight, avg batch 432 -> 256, 31.0 k -> 37.5 k nnEvals/s
    // (profiles/r03_steps/fibers.txt).
    // Round 4 measured the alternatives a larger max_batch_size invites (advisor, round 3), same benchmark with max_batch_size 1024,
    // A/B on one box whose device does 38.5 k (profiles/r04_steps/batcher/seal_rule_ab.txt): sealed at the LARGEST multiple only:
    // 34.4 of 43.3 k on another box (avg batch 492: with L leaves in flight, batches of L/2 leave the host's turnaround uncovered);
    // at any multiple while fewer than two batches are ahead, growing on otherwise: 36.2 k (avg 337); at any multiple while a slot of
    // the device is free: 37.0 k (avg 275: a batch that grows past a boundary must reach the NEXT one before it can go, and the
    // device drains meanwhile); at one granule always: 38.7 k, 100 % of the device (avg 248). So one granule it is, and engines and
    // staging are sized for that. KMX_BATCH_GROW_AHEAD=k restores growth: the largest multiple of the granule that max_batch_size
    // allows is the limit, and a batch is sealed at a smaller multiple only while fewer than k batches are launched or queued.
    // KMX_BATCH_QUANTUM overrides the granule (0 = off: only max_batch_size seals). include/katamx.h states this contract.
    {
      int q = -1;
      if(const char* e = getenv("KMX_BATCH_QUANTUM")) q = atoi(e);
      if(q < 0) {
        int dev = device;
        if(dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;  // "the current device" is what the engines below take as well
        hipDeviceProp_t prop;
        q = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
      }
      growAhead_ = 0;
      if(const char* e = getenv("KMX_BATCH_GROW_AHEAD")) growAhead_ = atoi(e) < 0 ? 0 : atoi(e);
      if(q > 0 && q < maxBatch) {
        sealAt_ = growAhead_ > 0 ? maxBatch / q * q : q;
        granule_ = q < sealAt_ ? q : 0;
      }
      else sealAt_ = maxBatch;
      if(const char* e = getenv("KMX_BATCH_LINGER_US")) lingerUs_ = atoi(e) < 0 ? 0 : atoi(e);
      if(const char* e = getenv("KMX_BATCH_LINGER_BIG_US")) lingerBigUs_ = atoi(e) < 0 ? 0 : atoi(e);
      // fault triage: one line per launched batch on stderr (slot, rows, what else is on the device) - the last lines before a device
      // fault name the batch sizes that were in flight (bench.py keeps them when the self-play leg dies)
      if(const char* e = getenv("KMX_BATCH_TRACE")) trace_ = atoi(e) != 0;
    }
    maxBatch = sealAt_;  // no batch ever holds more rows: engines and staging are sized for what can be used
    maxBatch_ = sealAt_;
    for(int i = 0; i < numSlots; i++) slots_.emplace_back();
    for(Slot& s : slots_) {
      s.eng.reset(new Engine(model, nnXLen, nnYLen, maxBatch, dtype, device));
      s.eng->setSharesDevice(maxInFlight > 1);
      s.sym.resize(maxBatch);
      s.opt.resize(maxBatch);
      s.rows.resize(maxBatch);
    }
    cin_ = slots_[0].eng->numInputChannels();
    gin_ = slots_[0].eng->numInputGlobalChannels();
    min_ = slots_[0].eng->numInputMetaChannels();
    if(cin_ > 32) throw Error(KMX_ERR_UNSUPPORTED, "batcher: more than 32 spatial input planes");
    dispatcher_ = std::thread([this] { dispatchLoop(); });
    completer_ = std::thread([this] { completeLoop(); });
  }
  ~Batcher() {
    {
      std::lock_guard<std::mutex> l(mu_);
      closing_ = true;
    }
    cvWork_.notify_all();
    cvComplete_.notify_all();
    cvFree_.notify_all();
    if(dispatcher_.joinable()) dispatcher_.join();
    if(completer_.joinable()) completer_.join();
  }

  int maxBatch() const { return maxBatch_; }  // rows per batch at most: the seal size (kmx_batcher_effective_batch)

  // exactly one of rowSpatial (fp32 NHWC planes, bit-packed here) and rowPacked (already in the staging layout) is given
  uint64_t submit(const float* rowSpatial, const unsigned char* rowPacked, const float* rowGlobal, const float* rowMeta, int symmetry, float optimism,
                  float* outPolicy, float* outValue, float* outScore, float* outOwnership) {
    if((!rowSpatial && !rowPacked) || !rowGlobal) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_submit: null row");
    if(!outPolicy || !outValue || !outScore) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_submit: null output");
    if(symmetry < 0 || symmetry > 7) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_submit: symmetry must be in 0..7");
    if((min_ > 0) != (rowMeta != nullptr))
      throw Error(KMX_ERR_INVALID_ARG, min_ > 0 ? "this net has an sgf-metadata encoder: rows need the metadata input"
                                                : "this net has no sgf-metadata encoder: the metadata input must be NULL");
    // fp32 planes are bit-packed BEFORE a row is reserved: a plane with a value other than 0 / 1 fails this call only, not the
    // batch its row would have shared with other callers' rows
    const size_t packedBytes = (size_t)slots_[0].eng->packedRowBytes();
    unsigned char localPacked[32 * 128];
    std::vector<unsigned char> bigPacked;
    unsigned char* packedHere = localPacked;
    if(!rowPacked) {
      if(packedBytes > sizeof(localPacked)) {
        bigPacked.resize(packedBytes);
        packedHere = bigPacked.data();
      }
      if(!packRowNHWC(rowSpatial, S_, cin_, packedHere))
        throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_submit: spatial features must be 0 or 1 (bit-packed staging); use kmx_eval for other inputs");
      rowPacked = packedHere;
    }
    int si, r;
    uint64_t ticket;
    {
      std::unique_lock<std::mutex> l(mu_);
      for(;;) {
        if(closing_) throw Error(KMX_ERR_INTERNAL, "kmx_batcher_submit: the batcher is shutting down");
        if(filling_ < 0) {
          int f = -1;
          for(size_t i = 0; i < slots_.size(); i++)
            if(slots_[i].state == FREE) { f = (int)i; break; }
          if(f < 0) {  // every staging set is filling, sealed or on the device
            cvFree_.wait(l);
            continue;
          }
          Slot& s = slots_[f];
          s.state = FILLING;
          s.count = 0;
          s.copied.store(0, std::memory_order_relaxed);
          s.anyOwner = false;
          filling_ = f;
        }
        Slot& s = slots_[filling_];
        si = filling_;
        r = s.count++;
        ticket = nextTicket_++;
        s.sym[r] = symmetry;
        s.opt[r] = optimism;
        s.anyOwner = s.anyOwner || outOwnership != nullptr;
        Pending& p = pending_[ticket];
        p.slot = si;
        s.rows[r] = RowOut{ticket, outPolicy, outValue, outScore, outOwnership};
        // full - or at a granule boundary with the device about to run dry: no further reservations, the next row opens a new batch
        if(s.count == sealAt_ || (granule_ > 0 && s.count % granule_ == 0 && running_ + (int)sealed_.size() < growAhead_)) {
          s.state = SEALED;
          sealed_.push_back(filling_);
          filling_ = -1;
        }
        break;
      }
    }
    cvWork_.notify_one();
    // stage the row outside the lock: the dispatcher launches only once every reserved row has been copied
    Slot& s = slots_[si];
    memcpy(s.eng->stagedPackedRow(r), rowPacked, packedBytes);
    memcpy(s.eng->stagedGlobalRow(r), rowGlobal, (size_t)gin_ * sizeof(float));
    if(min_ > 0) memcpy(s.eng->stagedMetaRow(r), rowMeta, (size_t)min_ * sizeof(float));
    s.copied.fetch_add(1, std::memory_order_release);
    return ticket;
  }

  void wait(uint64_t ticket) {
    int err;
    std::string msg;
    {
      std::unique_lock<std::mutex> l(mu_);
      auto it = pending_.find(ticket);
      if(it == pending_.end()) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_wait: unknown ticket (already waited for?)");
      while(!it->second.done) slots_[it->second.slot].cvDone.wait(l);  // (std::map: the element stays where it is while other tickets come and go)
      err = it->second.error;
      if(err != KMX_OK) msg = *it->second.message;
      pending_.erase(it);
    }
    if(err != KMX_OK) throw Error(err, msg);
  }

  void stats(uint64_t* rows, uint64_t* batches) {
    std::lock_guard<std::mutex> l(mu_);
    if(rows) *rows = rows_;
    if(batches) *batches = batches_;
  }
  int numInputMetaChannels() const { return min_; }

 private:
  enum State { FREE, FILLING, SEALED, RUNNING };
  struct RowOut {
    uint64_t ticket;
    float *policy, *value, *score, *ownership;
  };
  struct Slot {
    std::unique_ptr<Engine> eng;
    State state = FREE;
    int count = 0;                 // rows reserved
    std::atomic<int> copied{0};    // rows staged
    bool anyOwner = false;
    std::vector<int> sym;
    std::vector<float> opt;
    std::vector<RowOut> rows;      // where each row's results go
    int error = KMX_OK;
    std::string errorMsg;
    std::condition_variable cvDone;  // the waiters of the rows that were in this set
  };
  struct Pending {  // one per ticket, from submit to wait
    int slot = -1;
    bool done = false;
    int error = KMX_OK;
    std::shared_ptr<const std::string> message;  // shared by the rows of a failed batch
  };
  // finishes the rows of a set (lock held): marks their tickets, frees the set, wakes its waiters and one blocked submitter
  void finishSlot(Slot& s, int err, const std::string& msg) {
    std::shared_ptr<const std::string> m = err != KMX_OK ? std::make_shared<const std::string>(msg) : nullptr;
    for(int r = 0; r < s.count; r++) {
      auto it = pending_.find(s.rows[r].ticket);
      if(it == pending_.end()) continue;
      it->second.done = true;
      it->second.error = err;
      it->second.message = m;
    }
    s.state = FREE;
    s.cvDone.notify_all();
    cvFree_.notify_all();
  }

  // the helper threads of a device run on its NUMA node (numa.h): they touch every staged row and every result of its batches
  void bindToDeviceNode(const char* what) {
    const int dev = slots_[0].eng->device();
    numa::bindThisThreadToNode(numa::nodeOfDevice(dev), what, dev);
  }
  void dispatchLoop() {
    bindToDeviceNode("dispatcher");
    std::unique_lock<std::mutex> l(mu_);
    for(;;) {
      // A FULL batch goes as soon as fewer than max_in_flight are on the device. A partial batch goes only when the device
      // is idle: while a batch runs, arriving rows accumulate into the next one (sealing them at once - the reference's
      // greedy rule applied per in-flight slot - produces many small batches that share the device inefficiently; measured
      // with the reference's benchmark, 4 server threads: avg 45 rows per batch).
      // ... unless what is on the device is small: a pass over a few dozen rows occupies a fraction of the CUs (one work-group
      // per board and 32-96 channels) and is bound by its ~100 dependent launches, so a second small batch beside it is
      // nearly free - the regime of self-play with few game threads.
      cvWork_.wait(l, [&] {
        if(closing_ || (running_ < maxInFlight_ && !sealed_.empty())) return true;
        if(filling_ < 0 || slots_[filling_].count == 0) return false;
        return running_ == 0 || (running_ < maxInFlight_ && rowsOnDevice_ + slots_[filling_].count <= SMALL_ROWS);
      });
      // A short linger before a PARTIAL batch goes: the rows of the batch that has just been delivered come back one by one within a
      // few hundred microseconds (each waiter wakes, post-processes, descends again, featurises), and greedy sealing on the first of
      // them cuts what would be one batch into several small ones that then share the device - a pass costs ~2.5-3 ms whatever it
      // holds up to ~64 rows, so rows per pass is what counts (64 leaves in flight: 15.3 k rows/s with at most two batches in flight,
      // 11.9 k with eight). While fewer rows wait than the last delivered batch held, the dispatcher waits up to lingerUs_ more
      // (KMX_BATCH_LINGER_US, default 150; 0 = the reference's pure greedy rule, threadsafequeue.h:173-189) - at most once per batch,
      // 5 % of a pass.
      // Round 6: the linger is longer where batches are large. With ~256 leaves in flight (BASELINE configs[1] as written: `benchmark -t 256`;
      // self-play at 32 games x 8 leaves) two batches of ~115-135 rows alternate and each pass costs 4 ms: waiting up to lingerBigUs_ (default
      // 800 us; KMX_BATCH_LINGER_BIG_US) for a batch as large as the last one, once that held at least LINGER_BIG_ROWS rows, lets the batches
      // grow to 140-190 rows. Measured, two runs each, one box (profiles/r06_steps/midbatch/linger2.txt): `benchmark -v 1600 -t 256` 27.7 k
      // nnEvals/s without, 30.5 k at 600 us, 29.9 k at 1000 and 1500; self-play at 32 x 8 28.0 k NN rows/s without, 29.1 k at 600, 29.6 k at
      // 1000, 29.5 k at 1500; nothing changes at 8 x 8 (batches of ~27 rows never qualify: 22.0-22.6 k either way) nor on long searches with
      // 1024 leaves (batches seal full). With 100 games x 1 leaf (batches of ~42 rows, a few above 64) a threshold of 64 rows cost 6 %
      // (25.5 -> 24.0 k): the threshold is 96.
      const int lingerNow = lastBatchRows_ >= LINGER_BIG_ROWS ? std::max(lingerUs_, lingerBigUs_) : lingerUs_;
      if(!closing_ && sealed_.empty() && filling_ >= 0 && lingerNow > 0 && slots_[filling_].count < lastBatchRows_ &&
         slots_[filling_].count < sealAt_)
      {
        const int f = filling_;
        cvWork_.wait_for(l, std::chrono::microseconds(lingerNow), [&] {
          return closing_ || !sealed_.empty() || filling_ != f || slots_[f].count >= lastBatchRows_ || slots_[f].count >= sealAt_;
        });
        if(!closing_ && sealed_.empty() && filling_ < 0) continue;  // (cannot happen: only this thread seals a partial batch)
      }
      if(closing_) {
        // rows that were never launched: fail their waiters
        auto fail = [&](int i) { finishSlot(slots_[i], KMX_ERR_INTERNAL, "the batcher shut down before this row was evaluated"); };
        for(int i : sealed_) fail(i);
        sealed_.clear();
        if(filling_ >= 0) {
          fail(filling_);
          filling_ = -1;
        }
        return;
      }
      int si;
      if(!sealed_.empty()) {
        si = sealed_.front();
        sealed_.pop_front();
      }
      else {
        si = filling_;  // greedy: take what is there (threadsafequeue.h:173-189), the next row opens a new batch
        slots_[si].state = SEALED;
        filling_ = -1;
      }
      Slot& s = slots_[si];
      running_++;
      rowsOnDevice_ += s.count;
      const int n = s.count;
      const bool anyOwner = s.anyOwner;
      if(trace_) fprintf(stderr, "[kmx batch] slot %d rows %d beside %d rows in %d batches\n", si, n, rowsOnDevice_ - n, running_ - 1);
      l.unlock();
      while(s.copied.load(std::memory_order_acquire) < n) std::this_thread::yield();  // a row copy takes a few microseconds
      int err = KMX_OK;
      std::string msg;
      try {
        s.eng->launchStagedPacked(n, s.sym.data(), s.opt.data(), anyOwner);
      }
      catch(const Error& e) { err = e.code; msg = e.what(); }
      catch(const std::exception& e) { err = KMX_ERR_INTERNAL; msg = e.what(); }
      if(err != KMX_OK) {
        // a launch that failed half-way may have copies or kernels queued on this set's stream: nothing may reuse its pinned
        // staging before they have drained (best effort - the stream itself may be what failed)
        try { s.eng->sync(); } catch(...) {}
      }
      l.lock();
      s.error = err;
      s.errorMsg = msg;
      s.state = RUNNING;
      inflight_.push_back(si);
      cvComplete_.notify_one();
    }
  }

  void completeLoop() {
    bindToDeviceNode("completion");
    std::unique_lock<std::mutex> l(mu_);
    for(;;) {
      // a batch the dispatcher has counted (running_) but not yet handed over is still coming
      cvComplete_.wait(l, [&] { return !inflight_.empty() || (closing_ && running_ == 0); });
      if(inflight_.empty()) return;
      // Batches finish in launch order when they are large (each fills the device); small ones side by side (SMALL_ROWS) need not:
      // a batch behind the oldest that has already finished is delivered first instead of waiting for its elder. The streams are
      // queried OUTSIDE the lock (every submitter and the dispatcher need it; a query is a runtime call), on a snapshot of the
      // batches in flight; while none of several has finished the thread naps 30 us (or until the dispatcher hands over another
      // batch) and looks again, instead of blocking on the oldest while a younger one completes.
      size_t pick = 0;
      if(inflight_.size() > 1) {
        bool found = false;
        while(!found) {
          const std::vector<int> snap(inflight_.begin(), inflight_.end());
          l.unlock();
          int doneSlot = -1;
          for(int si : snap) {
            Slot& c = slots_[si];  // RUNNING: only this thread moves it on
            bool done = false;
            try { done = c.error != KMX_OK || c.eng->idle(); } catch(...) { done = true; }  // a failed batch is delivered (as failed) at once
            if(done) { doneSlot = si; break; }
          }
          l.lock();
          if(doneSlot >= 0) {
            for(size_t i = 0; i < inflight_.size(); i++)
              if(inflight_[i] == doneSlot) pick = i;
            found = true;
          }
          else if(inflight_.size() <= 1) { pick = 0; found = true; }  // (cannot shrink: only this thread removes; kept for clarity)
          else cvComplete_.wait_for(l, std::chrono::microseconds(30));
        }
      }
      const int si = inflight_[pick];
      inflight_.erase(inflight_.begin() + (long)pick);
      Slot& s = slots_[si];
      l.unlock();
      int err = s.error;
      std::string msg = s.errorMsg;
      if(err == KMX_OK) {
        try {
          s.eng->sync();
          // results -> the buffers the submitters named (they stay valid until kmx_batcher_wait returns)
          for(int r = 0; r < s.count; r++) {
            const RowOut& o = s.rows[r];
            memcpy(o.policy, s.eng->stagedPolicy(r), (size_t)(S_ + 1) * sizeof(float));
            memcpy(o.value, s.eng->stagedValue(r), 3 * sizeof(float));
            memcpy(o.score, s.eng->stagedScore(r), 6 * sizeof(float));
            if(o.ownership) memcpy(o.ownership, s.eng->stagedOwnership(r), (size_t)S_ * sizeof(float));
          }
        }
        catch(const Error& e) { err = e.code; msg = e.what(); }
        catch(const std::exception& e) { err = KMX_ERR_INTERNAL; msg = e.what(); }
      }
      l.lock();
      if(err == KMX_OK) {
        rows_ += (uint64_t)s.count;
        lastBatchRows_ = s.count;
        batches_ += 1;
      }
      rowsOnDevice_ -= s.count;
      finishSlot(s, err, msg);
      running_--;
      cvWork_.notify_one();
    }
  }

  int maxBatch_, maxInFlight_, sealAt_ = 0, granule_ = 0, growAhead_ = 0, S_ = 0, cin_ = 0, gin_ = 0, min_ = 0;
  std::deque<Slot> slots_;  // a Slot holds a condition variable: never moved
  std::mutex mu_;
  std::condition_variable cvWork_, cvComplete_, cvFree_;
  std::deque<int> sealed_, inflight_;
  int filling_ = -1, running_ = 0;
  int rowsOnDevice_ = 0;                    // rows of the batches between launch and completion
  int lastBatchRows_ = 0;                   // rows of the batch delivered last: what a partial batch is expected to grow to
  int lingerUs_ = 150;
  int lingerBigUs_ = 800;                   // ... once the last delivered batch held at least LINGER_BIG_ROWS rows
  static constexpr int LINGER_BIG_ROWS = 96;
  static constexpr int SMALL_ROWS = 96;     // partial batches may run side by side while the device holds at most this many rows
  bool closing_ = false;
  bool trace_ = false;
  uint64_t nextTicket_ = 1;
  // node-based map: references to its elements stay valid while other tickets come and go (a waiter sleeps holding one)
  std::map<uint64_t, Pending> pending_;
  uint64_t rows_ = 0, batches_ = 0;
  std::thread dispatcher_, completer_;
};

}  // namespace kmx

using namespace kmx;

struct kmx_batcher {
  std::unique_ptr<Batcher> b;
  int precision = KMX_PREC_BF16;
};

// shared with kmx_api.cpp
namespace kmx {
int apiGuarded(const std::function<void()>& f);
int apiDtypeFor(const kmx_context* ctx, const kmx_model* model);
const ModelDesc& apiModelDesc(const kmx_model* model);
void apiContextDims(const kmx_context* ctx, int* x, int* y);
int apiSetError(int code, const std::string& msg);
}  // namespace kmx

extern "C" {

int kmx_batcher_create(kmx_context* ctx, const kmx_model* model, int max_batch_size, int max_in_flight, int gpu_idx, kmx_batcher** out) {
  return apiGuarded([&] {
    if(!ctx || !model || !out) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_create: null argument");
    *out = nullptr;
    if(max_batch_size < 1 || max_batch_size > 65535) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_create: max_batch_size must be in 1..65535");
    if(max_in_flight < 1) max_in_flight = 2;
    if(max_in_flight > 8) max_in_flight = 8;
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error(KMX_ERR_DEVICE, "no usable HIP device (katamx has no CPU fallback)");
    const int dev = gpu_idx < 0 ? 0 : gpu_idx;
    if(dev >= ndev) throw Error(KMX_ERR_DEVICE, "kmx_batcher_create: device index out of range");
    int x, y;
    apiContextDims(ctx, &x, &y);
    std::unique_ptr<kmx_batcher> h(new kmx_batcher());
    // staging sets: the ones on the device, one filling, one being collected by its waiters
    const int dtype = apiDtypeFor(ctx, model);
    h->precision = dtype == DT_F16 ? KMX_PREC_FP16 : dtype == DT_F32 ? KMX_PREC_FP32 : KMX_PREC_BF16;
    h->b.reset(new Batcher(apiModelDesc(model), x, y, max_batch_size, dtype, dev, max_in_flight, max_in_flight + 2));
    *out = h.release();
  });
}
void kmx_batcher_free(kmx_batcher* b) { delete b; }

int kmx_batcher_submit(kmx_batcher* b, const float* row_spatial, const float* row_global, const float* row_meta, int symmetry,
                       float policy_optimism, float* out_policy, float* out_value, float* out_score, float* out_ownership, uint64_t* ticket) {
  return apiGuarded([&] {
    if(!b || !ticket) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_submit: null argument");
    if(!row_spatial) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_submit: null row");
    *ticket = b->b->submit(row_spatial, nullptr, row_global, row_meta, symmetry, policy_optimism, out_policy, out_value, out_score, out_ownership);
  });
}
int kmx_batcher_submit_packed(kmx_batcher* b, const uint8_t* row_packed, const float* row_global, const float* row_meta, int symmetry,
                              float policy_optimism, float* out_policy, float* out_value, float* out_score, float* out_ownership, uint64_t* ticket) {
  return apiGuarded([&] {
    if(!b || !ticket) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_submit_packed: null argument");
    if(!row_packed) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_submit_packed: null row");
    *ticket = b->b->submit(nullptr, row_packed, row_global, row_meta, symmetry, policy_optimism, out_policy, out_value, out_score, out_ownership);
  });
}
int kmx_batcher_wait(kmx_batcher* b, uint64_t ticket) {
  return apiGuarded([&] {
    if(!b) throw Error(KMX_ERR_INVALID_ARG, "kmx_batcher_wait: null batcher");
    b->b->wait(ticket);
  });
}
int kmx_batcher_precision(const kmx_batcher* b) { return b ? b->precision : KMX_PREC_AUTO; }
int kmx_batcher_effective_batch(const kmx_batcher* b) { return b ? b->b->maxBatch() : 0; }
int kmx_batcher_stats(kmx_batcher* b, uint64_t* rows, uint64_t* batches) {
  if(!b) return apiSetError(KMX_ERR_INVALID_ARG, "kmx_batcher_stats: null batcher");
  b->b->stats(rows, batches);
  return KMX_OK;
}

}  // extern "C"
