// katamx_leaf.h — the ticket side of the reference-side binding (integration/katamxbackend.cpp).
//
// `namespace NeuralNet` (cpp/neuralnet/nninterface.h:32-182) only has the synchronous batch call getOutput, which is
// why the reference needs server threads and a queue in front of it (NNEvaluator::serve, cpp/neuralnet/nneval.cpp:562-752).
// A caller that owns a leaf can instead hand it over itself and come back for the result: that is what a LeafPort offers,
// on top of the persistent leaf batcher of the C ABI (include/katamx.h, kmx_batcher_*). integration/katamx_nneval.cpp
// (this repo's NNEvaluator) is its caller; nothing in the unmodified reference needs this header.
//
// (Test builds link the binding against an implementation of the C ABI on the CPU oracle, oracle/kmx_abi_on_oracle.cpp, whose
// "batcher" evaluates each row synchronously inside submit: the evaluator's host logic then runs under the reference's own
// tests without a GPU. Nothing in this header or in the binding knows about that.)
#ifndef KATAMX_LEAF_H_
#define KATAMX_LEAF_H_

#include <cstdint>

#include "neuralnet/nninterface.h"

namespace KatamxLeaf {

struct Port;  // one per (model, device): a persistent leaf batcher

// maxBatchSize = rows per device batch; batchesInFlight <= 0 picks the backend default (2). gpuIdx < 0 = device 0.
Port* openPort(ComputeContext* context, const LoadedModel* loadedModel, Logger* logger, int maxBatchSize, int batchesInFlight, int gpuIdx);
void closePort(Port* port);  // no thread may be inside submit / wait
bool isUsingFP16(const Port* port);  // isUsingFP16 of nninterface.h:108 for this port's engines

// One row. rowSpatial: fp32 NHWC planes as NNInputs::fillRowV7 writes them (values 0 / 1; bit-packed by the callee straight
// into pinned staging), rowGlobal / rowMeta as NNResultBuf holds them (rowMeta NULL unless the net has a metadata encoder).
// The four output buffers (policy nnX*nnY+1, value 3, score 6, ownership nnX*nnY or NULL) must stay valid until wait()
// returns. Thread-safe; a thread may hold any number of tickets. Throws StringError.
uint64_t submit(
  Port* port, const float* rowSpatial, const float* rowGlobal, const float* rowMeta, int symmetry, float policyOptimism,
  float* outPolicy, float* outValue, float* outScore, float* outOwnership);
// The same for a row that was featurised as bit planes (integration/katamx_features.h; layout of kmx_eval_packed): numPlanes *
// ceil(nnX*nnY / 8) bytes, copied before the call returns.
uint64_t submitPacked(
  Port* port, const uint8_t* rowPacked, int numPlanes, const float* rowGlobal, const float* rowMeta, int symmetry, float policyOptimism,
  float* outPolicy, float* outValue, float* outScore, float* outOwnership);
void wait(Port* port, uint64_t ticket);  // blocks until the row's outputs are written; each ticket exactly once
void stats(Port* port, uint64_t& rows, uint64_t& batches);  // meaning of nneval.cpp:712-713

}  // namespace KatamxLeaf

#endif
