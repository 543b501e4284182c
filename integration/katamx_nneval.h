// katamx_nneval.h — ticket entry points of this repo's NNEvaluator (integration/katamx_nneval.cpp).
//
// integration/katamx_nneval.cpp is an implementation of the reference's class NNEvaluator (declared, unchanged, in
// cpp/neuralnet/nneval.h:82-298) that has NO server threads and no query queue: the thread that owns a position hashes it,
// looks it up in the cache, featurises it and hands the row straight to the device's persistent leaf batcher
// (katamx_leaf.h -> kmx_batcher_submit), then collects the result by ticket and post-processes it. It is linked INSTEAD OF
// cpp/neuralnet/nneval.cpp (oracle/Makefile: katago_hipx / katago_oraclex); every caller of the reference — Search,
// benchmark, selfplay, analysis, gtp, the tests — keeps calling NNEvaluator::evaluate and gets the same values.
//
// What the class interface cannot express is a caller that does not want to block: NNEvaluator::evaluate returns when
// the result is there (nneval.cpp:861-1262), which is why the reference's search needs one OS thread per leaf in flight
// (SURVEY 8f2). The two functions below split evaluate() at the hand-over:
//     KatamxNNEval::begin   hash -> cache -> featurise -> submit; returns at once (leaf.ready() if it was a cache hit)
//     KatamxNNEval::finish  wait for the ticket -> post-process -> cache store; the result is in leaf.buf->result
// so that a search thread can descend again (virtual losses applied) before it waits. begin + finish on the same thread
// is exactly evaluate(). The caller that keeps K leaves per OS thread in flight is the reference's own search, run on fibers
// (integration/katamx_fibers.h): evaluate() parks the calling fiber between begin and finish.
#ifndef KATAMX_NNEVAL_H_
#define KATAMX_NNEVAL_H_

#include <exception>
#include <memory>

#include "neuralnet/nneval.h"

namespace KatamxLeaf { struct Port; }

namespace KatamxNNEval {

// One position between begin() and finish(). The Board / BoardHistory / SGFMetadata the leaf was begun with and the
// NNResultBuf must stay alive and unchanged until finish() returns (post-processing reads legality from them, as
// nneval.cpp:960-974 does); a Leaf is not movable while in flight (the device writes into it).
struct Leaf {
  NNResultBuf* buf = NULL;
  const Board* board = NULL;
  const BoardHistory* history = NULL;
  Player nextPlayer = C_EMPTY;
  MiscNNInputParams nnInputParams;
  Hash128 nnHash;
  KatamxLeaf::Port* port = NULL;
  void* portSlot = NULL;   // the evaluator's bookkeeping of that port (rows in flight per device); owned by the evaluator
  uint64_t ticket = 0;
  bool inFlight = false;   // handed to the device, finish() must be called
  bool done = false;       // buf->result is final (cache hit, or finish() ran)
  bool skipCacheStore = false;
  bool ticketCollected = false;  // the ticket was waited for while the owner's fiber was parked (katamx_fibers.h); finish() must not wait again
  std::exception_ptr collectError;  // ... and this is what that wait threw
  std::shared_ptr<NNOutput> cachedWithoutOwnerMap;  // nneval.cpp:922-936: only the ownership map was missing
  float value[3];
  float score[6];

  bool ready() const { return done; }
  Leaf() {}
  Leaf(const Leaf&) = delete;
  Leaf& operator=(const Leaf&) = delete;
};

void begin(
  NNEvaluator& nnEval, const Board& board, const BoardHistory& history, Player nextPlayer, const SGFMetadata* sgfMeta,
  const MiscNNInputParams& nnInputParams, NNResultBuf& buf, bool skipCache, bool includeOwnerMap, Leaf& leaf);
void finish(NNEvaluator& nnEval, Leaf& leaf);

}  // namespace KatamxNNEval

#endif
