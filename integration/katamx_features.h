// katamx_features.h — featurisation of one position straight into the boundary's bit-plane row.
//
// The reference featurises into an fp32 row (NNInputs::fillRowV7, cpp/neuralnet/nninputs.cpp:2288-2731: 22 planes x nnX*nnY
// floats = 31 768 bytes at 19x19, zero-filled and then dotted with 1.0f) which its backends copy to the device as is. Every
// one of those 22 planes is 0/1, and the C ABI of this repository takes them as bits (include/katamx.h, kmx_eval_packed /
// kmx_batcher_submit_packed: the reference's own binaryInputNCHWPacked layout, cpp/dataio/trainingwrite.cpp:314-337). This is
// the featuriser that writes that form directly - 1 012 bytes per 19x19 row, no fp32 row in between:
//
//   packed[plane * planeBytes + (pos >> 3)] bit (7 - (pos & 7)),   pos = y * nnXLen + x,   planeBytes = ceil(nnXLen*nnYLen / 8)
//
// plus the 19 global features as floats. The meaning of every plane and global is the reference's (inputs version 7); the
// rules knowledge it needs - liberties, ladder reading, pass-alive area, ko and encore state, komi - is asked of the
// reference's Board / BoardHistory, which stay the owners of the game. tests/test_features_own.py holds it, bit for bit,
// to NNInputs::fillRowV7 over random games under every rule combination (integration/features_selftest.cpp).
#ifndef KATAMX_FEATURES_H_
#define KATAMX_FEATURES_H_

#include <cstddef>
#include <cstdint>

#include "game/board.h"
#include "game/boardhistory.h"
#include "neuralnet/nninputs.h"

namespace KatamxFeatures {

constexpr int NUM_PLANES_V7 = NNInputs::NUM_FEATURES_SPATIAL_V7;
constexpr int NUM_GLOBALS_V7 = NNInputs::NUM_FEATURES_GLOBAL_V7;

inline int planeBytes(int nnXLen, int nnYLen) { return (nnXLen * nnYLen + 7) / 8; }
inline size_t packedRowBytes(int nnXLen, int nnYLen) { return (size_t)NUM_PLANES_V7 * planeBytes(nnXLen, nnYLen); }
constexpr size_t MAX_PACKED_ROW_BYTES = (size_t)NUM_PLANES_V7 * ((NNPos::MAX_BOARD_AREA + 7) / 8);

// Inputs version 7 of one position, for the player to move. packed: packedRowBytes(nnXLen, nnYLen) bytes, rowGlobal:
// NUM_GLOBALS_V7 floats; both are overwritten entirely. Same arguments and preconditions as NNInputs::fillRowV7.
void fillPackedV7(
  const Board& board, const BoardHistory& hist, Player nextPlayer, const MiscNNInputParams& nnInputParams,
  int nnXLen, int nnYLen, uint8_t* packed, float* rowGlobal);

// The fp32 channels-last row (rowNHWC[pos * numPlanes + plane] = 0.0f / 1.0f) the bits stand for - what
// NNEvaluator::fillRowBufs owes its callers, and what the CPU-oracle build of the leaf port evaluates.
void unpackToNHWC(const uint8_t* packed, int nnXLen, int nnYLen, int numPlanes, float* rowNHWC);

}  // namespace KatamxFeatures

#endif
