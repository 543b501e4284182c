// katamx_nneval.cpp — this repo's implementation of the reference's class NNEvaluator (cpp/neuralnet/nneval.h:82-298),
// linked INSTEAD OF cpp/neuralnet/nneval.cpp (oracle/Makefile: katago_hipx, katago_oraclex). SURVEY 8 rows a1, a3, a22, f2.
//
// Shape. The reference's evaluator is two halves around a queue: clients featurise a position into an NNResultBuf, push it
// and sleep on a condition variable (nneval.cpp:861-958); server threads pop up to maxBatch buffers and call the backend's
// synchronous getOutput (nneval.cpp:562-752). Here there is no queue and there are no server threads. The thread that owns
// the position does everything that is host work - hash, cache, featurisation, legality, post-processing - and hands the row
// to the device's persistent leaf batcher (katamx_leaf.h), which packs it into pinned staging on this same thread, forms
// batches while the previous one computes, and delivers the logits into the NNOutput this thread allocated. One hop instead
// of two, no second wake-up, and the split begin()/finish() (katamx_nneval.h) lets a caller keep several leaves in flight.
//
// What is identical to the reference, and pinned by running its own tests on this evaluator (tests/test_nneval_own.py):
// the values. Same hash (NNInputs::getHash), the same feature rows (inputs version 7 from this repository's featuriser,
// integration/katamx_features.cpp, held bit for bit to NNInputs::fillRowV7; older versions from NNInputs::fillRowV3-6), same symmetry rule, and the same
// post-processing arithmetic in the same precision (nneval.cpp:960-1254): policy logits -> masked softmax (passing hack,
// dagger ban), value logits -> softmax in double, score / lead / variance-time / short-term-error transforms per model
// version, ownership tanh; the no-neural-net debugging mode draws its random outputs in the order nneval.cpp:612-673 does.
#include "katamx_nneval.h"

#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <set>
#include <shared_mutex>
#include <unordered_map>

#include "core/test.h"
#include "katamx_features.h"
#include "katamx_fibers.h"
#include "katamx_leaf.h"
#include "neuralnet/modelversion.h"

using namespace std;

// ---- small structs of nneval.h -------------------------------------------------------------------------------------
NNResultBuf::NNResultBuf()
  : hasResult(false), includeOwnerMap(false), boardXSizeForServer(0), boardYSizeForServer(0), hasRowMeta(false), result(nullptr),
    errorLogLockout(false), symmetry(NNInputs::SYMMETRY_NOTSPECIFIED), policyOptimism(0.0) {}
NNResultBuf::~NNResultBuf() {}

// Only the reference's server threads use these; kept so that code holding one still links.
NNServerBuf::NNServerBuf(const NNEvaluator& nnEval, const LoadedModel* model) : inputBuffers(NULL) {
  if(model != NULL)
    inputBuffers = NeuralNet::createInputBuffers(model, nnEval.getMaxBatchSize(), nnEval.getNNXLen(), nnEval.getNNYLen());
}
NNServerBuf::~NNServerBuf() {
  if(inputBuffers != NULL)
    NeuralNet::freeInputBuffers(inputBuffers);
}

// ---- the cache (nneval.h:19-44): direct-mapped on the low hash bits, one mutex of a pool per entry ----------------------
NNCacheTable::Entry::Entry() : ptr(nullptr) {}
NNCacheTable::Entry::~Entry() {}

NNCacheTable::NNCacheTable(int sizePowerOfTwo, int mutexPoolSizePowerOfTwo) {
  if(sizePowerOfTwo < 0 || sizePowerOfTwo > 63)
    throw StringError("NNCacheTable: Invalid sizePowerOfTwo: " + Global::intToString(sizePowerOfTwo));
  if(mutexPoolSizePowerOfTwo < 0 || mutexPoolSizePowerOfTwo > 31)
    throw StringError("NNCacheTable: Invalid mutexPoolSizePowerOfTwo: " + Global::intToString(mutexPoolSizePowerOfTwo));
  mutexPoolSizePowerOfTwo = std::min(mutexPoolSizePowerOfTwo, sizePowerOfTwo);
  tableSize = ((uint64_t)1) << sizePowerOfTwo;
  tableMask = tableSize - 1;
  entries = new Entry[tableSize];
  mutexPoolMask = (((uint32_t)1) << mutexPoolSizePowerOfTwo) - 1;
  mutexPool = new MutexPool(mutexPoolMask + 1);
}
NNCacheTable::~NNCacheTable() {
  delete[] entries;
  delete mutexPool;
}
bool NNCacheTable::get(Hash128 nnHash, shared_ptr<NNOutput>& ret) {
  ret.reset();  // whatever the caller still held is released before the lock is taken
  const uint64_t idx = nnHash.hash0 & tableMask;
  std::lock_guard<std::mutex> lock(mutexPool->getMutex((uint32_t)idx & mutexPoolMask));
  const shared_ptr<NNOutput>& slot = entries[idx].ptr;
  if(slot == nullptr || !(slot->nnHash == nnHash))
    return false;
  ret = slot;
  return true;
}
void NNCacheTable::set(const shared_ptr<NNOutput>& p) {
  shared_ptr<NNOutput> incoming(p);  // the evicted entry is destroyed after the lock is released
  const uint64_t idx = p->nnHash.hash0 & tableMask;
  {
    std::lock_guard<std::mutex> lock(mutexPool->getMutex((uint32_t)idx & mutexPoolMask));
    entries[idx].ptr.swap(incoming);
  }
}
void NNCacheTable::clear() {
  for(uint64_t idx = 0; idx < tableSize; idx++) {
    shared_ptr<NNOutput> evicted;
    {
      std::lock_guard<std::mutex> lock(mutexPool->getMutex((uint32_t)idx & mutexPoolMask));
      entries[idx].ptr.swap(evicted);
    }
  }
}

// ---- what begin()/finish() need of an evaluator, outside the class the reference header fixes -------------------------------
namespace {

struct PortSlot {
  KatamxLeaf::Port* port = NULL;
  int gpuIdx = 0;
  bool usingFP16 = false;
  std::mutex randMutex;
  std::unique_ptr<Rand> rand;  // seeded like the reference's server thread of the same index: the symmetry draws match
  std::atomic<int> rowsInFlight{0};  // submitted to this port's batcher and not yet collected (begin -> finish)
};

struct EvalState {
  int nnXLen = 0, nnYLen = 0, policySize = 0;
  int modelVersion = -1, inputsVersion = -1, numInputMetaChannels = 0;
  bool requireExactNNLen = false, debugSkipNeuralNet = false;
  ModelPostProcessParams post;
  NNCacheTable* cache = NULL;
  Logger* logger = NULL;
  string modelName, modelFileName;
  std::atomic<uint64_t>* cacheHits = NULL;
  std::atomic<bool>* doRandomize = NULL;
  std::atomic<int>* defaultSymmetry = NULL;
  // ports exist between spawnServerThreads and killServerThreads
  std::vector<std::unique_ptr<PortSlot>> ports;
  std::atomic<uint32_t> nextPort{0};
  bool fillFirst = false;  // KATAMX_PORT_POLICY=fill (pickPort)
  int fillTarget = 256;    // rows in flight at which a port is passed over under that policy
  // rows of the no-neural-net mode are produced under this lock, in arrival order, from one generator
  std::mutex nnlessMutex;
  std::unique_ptr<Rand> nnlessRand;
  std::atomic<uint64_t> nnlessRows{0};
  // waitForNextNNEvalIfAny bookkeeping (nneval.cpp:754-763)
  std::mutex flightMutex;
  std::condition_variable flightChanged;
  int inFlight = 0;
  uint64_t completions = 0;
  bool stopping = false;
};

std::shared_mutex registryMutex;
std::unordered_map<const NNEvaluator*, std::shared_ptr<EvalState>> registry;

std::shared_ptr<EvalState> stateOf(const NNEvaluator* e) {
  std::shared_lock<std::shared_mutex> lock(registryMutex);
  auto it = registry.find(e);
  if(it == registry.end())
    throw StringError("katamx NNEvaluator: unknown evaluator");
  return it->second;
}

void flightBegin(EvalState& st) {
  std::lock_guard<std::mutex> lock(st.flightMutex);
  st.inFlight++;
}
void flightEnd(EvalState& st) {
  {
    std::lock_guard<std::mutex> lock(st.flightMutex);
    st.inFlight--;
    st.completions++;
  }
  st.flightChanged.notify_all();
}

double softPlusD(double x) {
  return x > 40.0 ? x : log(1.0 + exp(x));  // linear above 40, as nneval.cpp:766-772
}

// The position the opening "dagger" pattern bans (nneval.cpp:774-810): a 9 x 8 corner template, tried under 8 symmetries.
// Template cells: '.' must be empty, 'o' own stone, 'x' opponent stone, 'B' the banned point (any content).
const char* const DAGGER_ROWS[9] = {"........", "........", "..xo....", "..xo....", "........", ".xo.....", ".B......", "........", "........"};
bool daggerBan(const Board& board, Player pla, int symmetry, Loc& banned) {
  banned = Board::NULL_LOC;
  for(int ty = 0; ty < 9; ty++)
    for(int tx = 0; tx < 8; tx++) {
      int x = tx, y = ty;
      if(symmetry & 1) std::swap(x, y);
      if(symmetry & 2) x = board.x_size - 1 - x;
      if(symmetry & 4) y = board.y_size - 1 - y;
      const Loc loc = Location::getLoc(x, y, board.x_size);
      const char want = DAGGER_ROWS[ty][tx];
      const Color c = board.colors[loc];
      if(want == '.' && c != C_EMPTY) return false;
      if(want == 'o' && c != pla) return false;
      if(want == 'x' && c != getOpp(pla)) return false;
      if(want == 'B') banned = loc;
    }
  return true;
}

// ---- post-processing (nneval.cpp:960-1254): logits of the side to move -> probabilities and scores from White's side ----
void policyToProbabilities(EvalState& st, const Board& board, const BoardHistory& history, Player pla, const MiscNNInputParams& params, NNResultBuf& buf) {
  NNOutput& out = *buf.result;
  float* const policy = out.policyProbs;
  const int n = st.policySize;
  const float scaling = st.post.outputScaleMultiplier / params.nnPolicyTemperature;
  bool legal[NNPos::MAX_NN_POLICY_SIZE];
  testAssert(pla == history.presumedNextMovePla);
  for(int pos = 0; pos < n; pos++)
    legal[pos] = history.isLegal(board, NNPos::posToLoc(pos, board.x_size, board.y_size, st.nnXLen, st.nnYLen), pla);
  if(params.avoidMYTDaggerHack && board.x_size >= 13 && board.y_size >= 13)
    for(int symmetry = 0; symmetry < 8; symmetry++) {
      Loc banned;
      if(daggerBan(board, pla, symmetry, banned) && banned != Board::NULL_LOC)
        legal[NNPos::locToPos(banned, board.x_size, st.nnXLen, st.nnYLen)] = false;
    }
  int numLegal = 0;
  float top = -1e25f;
  for(int pos = 0; pos < n; pos++) {
    const float v = legal[pos] ? policy[pos] * scaling : -1e30f;
    numLegal += legal[pos] ? 1 : 0;
    policy[pos] = v;
    if(v > top) top = v;
  }
  testAssert(numLegal > 0);
  float sum = 0.0f;
  const int passPos = NNPos::locToPos(Board::PASS_LOC, board.x_size, st.nnXLen, st.nnYLen);
  testAssert(passPos == n - 1);
  for(int pos = 0; pos < n; pos++) {
    float e = std::exp(policy[pos] - top);
    if(params.enablePassingHacks && pos == passPos)  // passing prior capped at 19x everything else, floored at 1e-20
      e = std::max(1e-20f, std::min(e, sum * 19.0f));
    policy[pos] = e;
    sum += e;
  }
  if(!std::isfinite(sum)) {
    cout << "Got nonfinite for policy sum" << endl;
    history.printDebugInfo(cout, board);
    throw StringError("Got nonfinite for policy sum");
  }
  if(sum <= 0.0) {  // every legal move rounded to zero: fall back to uniform
    if(!buf.errorLogLockout && st.logger != NULL) {
      buf.errorLogLockout = true;
      st.logger->write("Warning: all legal moves rounded to 0 probability for " + st.modelFileName);
    }
    const float uniform = 1.0f / numLegal;
    for(int pos = 0; pos < n; pos++) policy[pos] = legal[pos] ? uniform : -1.0f;
  }
  else
    for(int pos = 0; pos < n; pos++) policy[pos] = legal[pos] ? (policy[pos] / sum) : -1.0f;
  for(int pos = n; pos < NNPos::MAX_NN_POLICY_SIZE; pos++) policy[pos] = -1.0f;
  out.policyOptimismUsed = (float)params.policyOptimism;
}

struct Softmax3 {
  double win, loss, noResult, sum;
};
Softmax3 softmax3(double winLogit, double lossLogit, double noResultLogit, bool noResultImpossible) {
  const double top = std::max(std::max(winLogit, lossLogit), noResultLogit);
  Softmax3 s;
  s.win = exp(winLogit - top);
  s.loss = exp(lossLogit - top);
  s.noResult = noResultImpossible ? 0.0 : exp(noResultLogit - top);
  s.sum = s.win + s.loss + s.noResult;
  s.win /= s.sum;
  s.loss /= s.sum;
  s.noResult /= s.sum;
  return s;
}

void valueToWhitePerspective(EvalState& st, const Board& board, const BoardHistory& history, Player pla, NNOutput& out) {
  const double k = st.post.outputScaleMultiplier;
  const double winLogit = out.whiteWinProb * k, lossLogit = out.whiteLossProb * k;
  double noResultLogit = out.whiteNoResultProb * k;
  const bool white = pla == P_WHITE;
  if(st.modelVersion == 3) {
    // version 3 nets put a pre-arctan score value where the score mean goes
    const double scoreValue = atan(out.whiteScoreMean * k) * 0.63661977236758134308;
    const Softmax3 p = softmax3(winLogit, lossLogit, noResultLogit, false);
    if(!std::isfinite(p.sum) || !std::isfinite(scoreValue)) {
      cout << "Got nonfinite for nneval value" << endl;
      cout << winLogit << " " << lossLogit << " " << noResultLogit << " " << scoreValue << endl;
      throw StringError("Got nonfinite for nneval value");
    }
    const float score = (float)ScoreValue::approxWhiteScoreOfScoreValueSmooth(scoreValue, 0.0, 2.0, board.sqrtBoardArea());
    out.whiteWinProb = (float)(white ? p.win : p.loss);
    out.whiteLossProb = (float)(white ? p.loss : p.win);
    out.whiteNoResultProb = (float)p.noResult;
    out.whiteScoreMean = white ? score : -score;
    out.whiteScoreMeanSq = out.whiteScoreMean * out.whiteScoreMean;
    out.whiteLead = out.whiteScoreMean;
    out.varTimeLeft = -1;
    out.shorttermWinlossError = -1;
    out.shorttermScoreError = -1;
    return;
  }
  if(st.modelVersion < 4)
    throw StringError("NNEval value postprocessing not implemented for model version");

  const double scoreMeanRaw = out.whiteScoreMean * k, scoreStdevRaw = out.whiteScoreMeanSq * k, leadRaw = out.whiteLead * k;
  const double varTimeRaw = out.varTimeLeft * k, stWinlossRaw = out.shorttermWinlossError * k, stScoreRaw = out.shorttermScoreError * k;
  // without a simple ko rule or territory scoring a game cannot end without a result
  const bool noResultImpossible = history.rules.koRule != Rules::KO_SIMPLE && history.rules.scoringRule != Rules::SCORING_TERRITORY;
  if(noResultImpossible)
    noResultLogit -= 100000.0;
  const Softmax3 p = softmax3(winLogit, lossLogit, noResultLogit, noResultImpossible);

  double scoreMean = scoreMeanRaw * st.post.scoreMeanMultiplier;
  const double scoreStdev = softPlusD(scoreStdevRaw) * st.post.scoreStdevMultiplier;
  double scoreMeanSq = scoreMean * scoreMean + scoreStdev * scoreStdev;
  double lead = leadRaw * st.post.leadMultiplier;
  const double varTimeLeft = softPlusD(varTimeRaw) * st.post.varianceTimeMultiplier;
  // the net's score outputs are conditional on a result; no-result counts as score 0
  const double withResult = 1.0 - p.noResult;
  scoreMean *= withResult;
  scoreMeanSq *= withResult;
  lead *= withResult;

  double stWinloss, stScore;
  if(st.modelVersion >= 14) {
    const double a = softPlusD(stWinlossRaw * 0.5), b = softPlusD(stScoreRaw * 0.5);
    stWinloss = sqrt(a * a * st.post.shorttermValueErrorMultiplier);
    stScore = sqrt(b * b * st.post.shorttermScoreErrorMultiplier);
  }
  else if(st.modelVersion >= 10) {
    stWinloss = sqrt(softPlusD(stWinlossRaw) * st.post.shorttermValueErrorMultiplier);
    stScore = sqrt(softPlusD(stScoreRaw) * st.post.shorttermScoreErrorMultiplier);
  }
  else {
    stWinloss = softPlusD(stWinlossRaw);
    stScore = softPlusD(stScoreRaw) * 10.0;
  }
  if(!std::isfinite(p.sum) || !std::isfinite(scoreMean) || !std::isfinite(scoreMeanSq) || !std::isfinite(lead) || !std::isfinite(varTimeLeft) ||
     !std::isfinite(stWinloss) || !std::isfinite(stScore)) {
    cout << "Got nonfinite for nneval value" << endl;
    cout << winLogit << " " << lossLogit << " " << noResultLogit << " " << scoreMean << " " << scoreMeanSq << " " << lead << " " << varTimeLeft
         << " " << stWinloss << " " << stScore << endl;
    throw StringError("Got nonfinite for nneval value");
  }
  out.whiteWinProb = (float)(white ? p.win : p.loss);
  out.whiteLossProb = (float)(white ? p.loss : p.win);
  out.whiteNoResultProb = (float)p.noResult;
  out.whiteScoreMean = white ? (float)scoreMean : -(float)scoreMean;
  out.whiteScoreMeanSq = (float)scoreMeanSq;
  out.whiteLead = white ? (float)lead : -(float)lead;
  const bool hasErrorHeads = st.modelVersion >= 9;
  out.varTimeLeft = hasErrorHeads ? (float)varTimeLeft : -1;
  out.shorttermWinlossError = hasErrorHeads ? (float)stWinloss : -1;
  out.shorttermScoreError = hasErrorHeads ? (float)stScore : -1;
}

void ownershipToWhitePerspective(EvalState& st, const Board& board, Player pla, NNOutput& out) {
  if(out.whiteOwnerMap == NULL)
    return;
  if(st.modelVersion < 3)
    throw StringError("NNEval value postprocessing not implemented for model version");
  const float k = st.post.outputScaleMultiplier;
  for(int y = 0; y < st.nnYLen; y++)
    for(int x = 0; x < st.nnXLen; x++) {
      float& o = out.whiteOwnerMap[y * st.nnXLen + x];
      if(y >= board.y_size || x >= board.x_size) o = 0.0f;
      else o = pla == P_WHITE ? tanh(o * k) : -tanh(o * k);
    }
}

// The debugging mode without a net (nneval.cpp:612-673): unnormalised random logits, drawn in this order per row.
void fillRandomOutput(EvalState& st, NNResultBuf& buf, int boardXSize, int boardYSize) {
  std::lock_guard<std::mutex> lock(st.nnlessMutex);
  Rand& rand = *st.nnlessRand;
  NNOutput& out = *buf.result;
  std::fill(out.policyProbs, out.policyProbs + NNPos::MAX_NN_POLICY_SIZE, 0.0f);
  for(int y = 0; y < boardYSize; y++)
    for(int x = 0; x < boardXSize; x++) out.policyProbs[NNPos::xyToPos(x, y, st.nnXLen)] = (float)rand.nextGaussian();
  out.policyProbs[NNPos::locToPos(Board::PASS_LOC, boardXSize, st.nnXLen, st.nnYLen)] = (float)rand.nextGaussian();
  if(out.whiteOwnerMap != NULL) {
    std::fill(out.whiteOwnerMap, out.whiteOwnerMap + st.nnXLen * st.nnYLen, 0.0f);
    for(int y = 0; y < boardYSize; y++)
      for(int x = 0; x < boardXSize; x++) out.whiteOwnerMap[NNPos::xyToPos(x, y, st.nnXLen)] = (float)rand.nextGaussian() * 0.20f;
  }
  const double win = 0.0 + rand.nextGaussian() * 0.20, loss = 0.0 + rand.nextGaussian() * 0.20;
  const double scoreMean = 0.0 + rand.nextGaussian() * 0.20, scoreMeanSq = 0.0 + rand.nextGaussian() * 0.20;
  const double noResult = 0.0 + rand.nextGaussian() * 0.20;
  out.whiteWinProb = (float)win;
  out.whiteLossProb = (float)loss;
  out.whiteNoResultProb = (float)noResult;
  out.whiteScoreMean = (float)scoreMean;
  out.whiteScoreMeanSq = (float)scoreMeanSq;
  out.whiteLead = (float)scoreMean;
  out.varTimeLeft = (float)(0.5 * boardXSize * boardYSize);
  out.shorttermWinlossError = 0.0f;
  out.shorttermScoreError = 0.0f;
  out.policyOptimismUsed = (float)buf.policyOptimism;
  st.nnlessRows.fetch_add(1, std::memory_order_relaxed);
}

// Featurisation (row a2). Inputs version 7 - every net since model version 8 - is written straight into the boundary's
// bit-plane row by this repository's featuriser (katamx_features.h: 1 012 bytes per 19x19 row instead of the 31 768-byte fp32 row
// of NNInputs::fillRowV7, ladder planes of the parent positions looked up instead of read again); older input versions keep the
// reference's functions and cross the boundary as fp32 rows. KATAMX_FEATURES = own (default) | reference (fillRowV7 + fp32 rows,
// for A/B) | check (both, any difference is fatal: tests/test_nneval_own.py runs the reference's commands in this mode).
enum FeatureMode { FEATURES_OWN = 0, FEATURES_REFERENCE = 1, FEATURES_CHECK = 2 };
FeatureMode featureMode() {
  static const FeatureMode mode = [] {
    const char* env = getenv("KATAMX_FEATURES");
    if(env == NULL || strcmp(env, "own") == 0) return FEATURES_OWN;
    if(strcmp(env, "reference") == 0) return FEATURES_REFERENCE;
    if(strcmp(env, "check") == 0) return FEATURES_CHECK;
    throw StringError("KATAMX_FEATURES must be own, reference or check");
  }();
  return mode;
}

void metadataRow(EvalState& st, const Board& board, Player pla, const SGFMetadata* sgfMeta, NNResultBuf& buf) {
  if(buf.rowMetaBuf.size() < (size_t)st.numInputMetaChannels) buf.rowMetaBuf.resize(st.numInputMetaChannels);
  buf.hasRowMeta = st.numInputMetaChannels > 0;
  if(buf.hasRowMeta) {
    if(sgfMeta == NULL)
      Global::fatalError("SGFMetadata is required for " + st.modelName + " but was not provided");
    if(!sgfMeta->initialized)
      Global::fatalError("SGFMetadata is required for " + st.modelName + " but was not initialized. Did you specify humanSLProfile=... in katago's config or via overrides?");
    SGFMetadata::fillMetadataRow(sgfMeta, buf.rowMetaBuf.data(), pla, board.x_size * board.y_size);
  }
}

// The reference's featurisers: fp32 planes into buf.rowSpatialBuf, globals into buf.rowGlobalBuf.
void referenceRow(EvalState& st, const Board& board, const BoardHistory& history, Player pla, const MiscNNInputParams& params, float* sp, float* gl) {
  // rows cross the boundary channels-last whatever the evaluator was constructed with (katamx.h conventions)
  const bool nhwc = true;
  static_assert(NNModelVersion::latestInputsVersionImplemented == 7, "a new inputs version needs a case here");
  switch(st.inputsVersion) {
    case 3: NNInputs::fillRowV3(board, history, pla, params, st.nnXLen, st.nnYLen, nhwc, sp, gl); break;
    case 4: NNInputs::fillRowV4(board, history, pla, params, st.nnXLen, st.nnYLen, nhwc, sp, gl); break;
    case 5: NNInputs::fillRowV5(board, history, pla, params, st.nnXLen, st.nnYLen, nhwc, sp, gl); break;
    case 6: NNInputs::fillRowV6(board, history, pla, params, st.nnXLen, st.nnYLen, nhwc, sp, gl); break;
    case 7: NNInputs::fillRowV7(board, history, pla, params, st.nnXLen, st.nnYLen, nhwc, sp, gl); break;
    default: ASSERT_UNREACHABLE;
  }
}

// Fills buf's global (and metadata) row and EITHER `packed` (returns true; KatamxFeatures::MAX_PACKED_ROW_BYTES) OR
// buf.rowSpatialBuf (returns false).
bool featurise(EvalState& st, const Board& board, const BoardHistory& history, Player pla, const SGFMetadata* sgfMeta, const MiscNNInputParams& params,
               NNResultBuf& buf, uint8_t* packed) {
  const int numPlanes = NNModelVersion::getNumSpatialFeatures(st.modelVersion);
  const size_t spatialLen = (size_t)numPlanes * st.nnXLen * st.nnYLen;
  const size_t globalLen = (size_t)NNModelVersion::getNumGlobalFeatures(st.modelVersion);
  if(buf.rowGlobalBuf.size() < globalLen) buf.rowGlobalBuf.resize(globalLen);
  metadataRow(st, board, pla, sgfMeta, buf);
  const FeatureMode mode = featureMode();
  const bool own = st.inputsVersion == 7 && mode != FEATURES_REFERENCE && packed != NULL;
  if(own)
    KatamxFeatures::fillPackedV7(board, history, pla, params, st.nnXLen, st.nnYLen, packed, buf.rowGlobalBuf.data());
  if(!own || mode == FEATURES_CHECK) {
    if(buf.rowSpatialBuf.size() < spatialLen) buf.rowSpatialBuf.resize(spatialLen);
    std::vector<float> refGlobal(own ? globalLen : 0);
    referenceRow(st, board, history, pla, params, buf.rowSpatialBuf.data(), own ? refGlobal.data() : buf.rowGlobalBuf.data());
    if(own) {
      std::vector<float> mine(spatialLen);
      KatamxFeatures::unpackToNHWC(packed, st.nnXLen, st.nnYLen, numPlanes, mine.data());
      if(memcmp(mine.data(), buf.rowSpatialBuf.data(), spatialLen * sizeof(float)) != 0 ||
         memcmp(refGlobal.data(), buf.rowGlobalBuf.data(), globalLen * sizeof(float)) != 0)
        Global::fatalError("KATAMX_FEATURES=check: KatamxFeatures::fillPackedV7 and NNInputs::fillRowV7 differ on\n" + Board::toStringSimple(board, '\n'));
    }
  }
  return own;
}

// Which device gets this row (one port per server-thread index of the reference's configuration, i.e. per GPU: nneval.cpp:399-407).
// The reference's server threads pull from ONE queue, so an idle GPU takes the next rows whatever the others are doing; the
// equivalent for rows that are pushed is the port with the FEWEST rows in flight. (Round 3 dealt rows round-robin: with cache
// hits, games ending and searches of different length the ports drift apart, and a pass costs the same for 40 rows as for 60 -
// the lighter device idles while the heavier one queues a second batch.) Ties go round: the scan starts one port further each time,
// which with equal loads is exactly the round-robin order. Spreading, not filling one device first: a pass over L/N rows is
// shorter than a pass over L, so N devices with L/N rows each finish L rows sooner than L/256 devices with full batches.
// KATAMX_PORT_POLICY = spread (default) | fill (round 5): least-loaded-first hands every device an equal share of
// the leaves in flight - at 8 GPUs x 8 games each device then sees batches well below its granule, where a pass costs the same for 20
// rows as for 60. `fill` is the alternative to A/B on a node: ports are taken in their fixed order and a port is passed over only once
// it holds `fillTarget` rows (one device granule, KATAMX_PORT_FILL_ROWS overrides); when every port is at its target the least loaded
// takes the row. With one port both policies are the same code path. tests/test_schedule_dryrun.py runs both on eight fake devices.
PortSlot& pickPort(EvalState& st) {
  const size_t n = st.ports.size();
  size_t best = 0;
  if(n > 1 && st.fillFirst) {
    size_t k = 0;
    while(k < n && st.ports[k]->rowsInFlight.load(std::memory_order_relaxed) >= st.fillTarget) k++;
    if(k < n) {
      st.ports[k]->rowsInFlight.fetch_add(1, std::memory_order_relaxed);
      return *st.ports[k];
    }
  }
  const size_t start = st.nextPort.fetch_add(1, std::memory_order_relaxed) % n;
  best = start;
  if(n > 1) {
    int bestLoad = st.ports[start]->rowsInFlight.load(std::memory_order_relaxed);
    for(size_t k = 1; k < n && bestLoad > 0; k++) {
      const size_t i = (start + k) % n;
      const int load = st.ports[i]->rowsInFlight.load(std::memory_order_relaxed);
      if(load < bestLoad) {
        bestLoad = load;
        best = i;
      }
    }
  }
  st.ports[best]->rowsInFlight.fetch_add(1, std::memory_order_relaxed);
  return *st.ports[best];
}

}  // namespace

// ---- begin / finish -----------------------------------------------------------------------------------------------------
void KatamxNNEval::begin(
  NNEvaluator& nnEval, const Board& board, const BoardHistory& history, Player nextPlayer, const SGFMetadata* sgfMeta,
  const MiscNNInputParams& nnInputParamsArg, NNResultBuf& buf, bool skipCache, bool includeOwnerMap, Leaf& leaf
) {
  const std::shared_ptr<EvalState> sp = stateOf(&nnEval);
  EvalState& st = *sp;
  testAssert(!st.stopping);
  testAssert(!leaf.inFlight);
  buf.hasResult = false;
  if(board.x_size > st.nnXLen || board.y_size > st.nnYLen)
    throw StringError("NNEvaluator was configured with nnXLen = " + Global::intToString(st.nnXLen) + " nnYLen = " + Global::intToString(st.nnYLen) +
                      " but was asked to evaluate board with larger x or y size");
  if(st.requireExactNNLen && (board.x_size != st.nnXLen || board.y_size != st.nnYLen))
    throw StringError("NNEvaluator was configured with nnXLen = " + Global::intToString(st.nnXLen) + " nnYLen = " + Global::intToString(st.nnYLen) +
                      " and requireExactNNLen, but was asked to evaluate board with different x or y size");

  leaf.buf = &buf;
  leaf.board = &board;
  leaf.history = &history;
  leaf.nextPlayer = nextPlayer;
  leaf.nnInputParams = nnInputParamsArg;
  leaf.done = false;
  leaf.cachedWithoutOwnerMap.reset();
  if(st.numInputMetaChannels > 0)
    leaf.nnInputParams.policyOptimism = 0.0;  // nets conditioned on sgf metadata are evaluated without policy optimism
  const MiscNNInputParams& params = leaf.nnInputParams;

  leaf.nnHash = NNInputs::getHash(board, history, nextPlayer, params);
  if(st.numInputMetaChannels > 0) {
    if(sgfMeta == NULL)
      Global::fatalError("SGFMetadata is required for " + st.modelName + " but was not provided");
    if(!sgfMeta->initialized)
      Global::fatalError("SGFMetadata is required for " + st.modelName + " but was not initialized. Did you specify humanSLProfile=... in katago's config or via overrides?");
    leaf.nnHash ^= sgfMeta->getHash(nextPlayer);
  }

  if(st.cache != NULL && !skipCache && st.cache->get(leaf.nnHash, buf.result)) {
    if(!includeOwnerMap || buf.result->whiteOwnerMap != NULL) {
      st.cacheHits->fetch_add(1, std::memory_order_relaxed);
      buf.hasResult = true;
      leaf.done = true;
      return;
    }
    // cached, but without the ownership map that is wanted now: evaluate again for the map only (finish() keeps the cached
    // policy and values, so that a different random symmetry does not perturb a search that already used them)
    leaf.cachedWithoutOwnerMap = std::move(buf.result);
    buf.result = nullptr;
  }
  buf.includeOwnerMap = includeOwnerMap;
  buf.boardXSizeForServer = board.x_size;
  buf.boardYSizeForServer = board.y_size;
  buf.symmetry = params.symmetry;
  buf.policyOptimism = params.policyOptimism;

  // the output this thread owns; the device writes the policy (and ownership) logits straight into it
  std::shared_ptr<NNOutput> out = std::make_shared<NNOutput>();
  out->nnXLen = st.nnXLen;
  out->nnYLen = st.nnYLen;
  out->whiteOwnerMap = includeOwnerMap ? new float[st.nnXLen * st.nnYLen] : NULL;
  buf.result = out;

  flightBegin(st);
  if(st.debugSkipNeuralNet) {
    fillRandomOutput(st, buf, board.x_size, board.y_size);
    leaf.port = NULL;
    leaf.inFlight = true;
    return;
  }
  try {
    uint8_t packedRow[KatamxFeatures::MAX_PACKED_ROW_BYTES];
    const bool rowIsPacked = featurise(st, board, history, nextPlayer, sgfMeta, params, buf, packedRow);
    if(st.ports.empty())
      throw StringError("NNEvaluator::evaluate called before spawnServerThreads");
    PortSlot& slot = pickPort(st);
    leaf.portSlot = &slot;
    if(buf.symmetry == NNInputs::SYMMETRY_NOTSPECIFIED) {
      if(st.doRandomize->load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(slot.randMutex);
        buf.symmetry = (int)slot.rand->nextUInt(SymmetryHelpers::NUM_SYMMETRIES);
      }
      else {
        buf.symmetry = st.defaultSymmetry->load(std::memory_order_acquire);
        testAssert(buf.symmetry >= 0 && buf.symmetry <= SymmetryHelpers::NUM_SYMMETRIES - 1);
      }
    }
    leaf.port = slot.port;
    const float* rowMeta = buf.hasRowMeta ? buf.rowMetaBuf.data() : NULL;
    if(rowIsPacked)
      leaf.ticket = KatamxLeaf::submitPacked(
        slot.port, packedRow, KatamxFeatures::NUM_PLANES_V7, buf.rowGlobalBuf.data(), rowMeta, buf.symmetry, (float)buf.policyOptimism,
        out->policyProbs, leaf.value, leaf.score, out->whiteOwnerMap);
    else
      leaf.ticket = KatamxLeaf::submit(
        slot.port, buf.rowSpatialBuf.data(), buf.rowGlobalBuf.data(), rowMeta, buf.symmetry, (float)buf.policyOptimism, out->policyProbs,
        leaf.value, leaf.score, out->whiteOwnerMap);
    leaf.inFlight = true;
  }
  catch(...) {
    if(leaf.portSlot != NULL) {
      static_cast<PortSlot*>(leaf.portSlot)->rowsInFlight.fetch_sub(1, std::memory_order_relaxed);
      leaf.portSlot = NULL;
    }
    flightEnd(st);
    throw;
  }
}

void KatamxNNEval::finish(NNEvaluator& nnEval, Leaf& leaf) {
  if(leaf.done)
    return;
  testAssert(leaf.inFlight);
  const std::shared_ptr<EvalState> sp = stateOf(&nnEval);
  EvalState& st = *sp;
  NNResultBuf& buf = *leaf.buf;
  leaf.inFlight = false;
  if(leaf.portSlot != NULL) {
    static_cast<PortSlot*>(leaf.portSlot)->rowsInFlight.fetch_sub(1, std::memory_order_relaxed);
    leaf.portSlot = NULL;
  }
  if(leaf.port != NULL) {
    try {
      if(!leaf.ticketCollected)
        KatamxLeaf::wait(leaf.port, leaf.ticket);
      else if(leaf.collectError)
        std::rethrow_exception(leaf.collectError);
    }
    catch(...) {
      flightEnd(st);
      throw;
    }
    // logits -> the NNOutput fields the post-processing reads them from (field order of the backends, eigenbackend.cpp:2569-2626)
    NNOutput& out = *buf.result;
    out.whiteWinProb = leaf.value[0];
    out.whiteLossProb = leaf.value[1];
    out.whiteNoResultProb = leaf.value[2];
    out.whiteScoreMean = leaf.score[0];
    out.whiteScoreMeanSq = leaf.score[1];
    out.whiteLead = leaf.score[2];
    out.varTimeLeft = leaf.score[3];
    out.shorttermWinlossError = st.modelVersion >= 9 ? leaf.score[4] : 0;
    out.shorttermScoreError = st.modelVersion >= 9 ? leaf.score[5] : 0;
  }
  flightEnd(st);
  buf.hasResult = true;

  NNOutput& out = *buf.result;
  if(leaf.cachedWithoutOwnerMap != nullptr) {
    const NNOutput& old = *leaf.cachedWithoutOwnerMap;
    testAssert(out.whiteOwnerMap != NULL);
    out.whiteWinProb = old.whiteWinProb;
    out.whiteLossProb = old.whiteLossProb;
    out.whiteNoResultProb = old.whiteNoResultProb;
    out.whiteScoreMean = old.whiteScoreMean;
    out.whiteScoreMeanSq = old.whiteScoreMeanSq;
    out.whiteLead = old.whiteLead;
    out.varTimeLeft = old.varTimeLeft;
    out.shorttermWinlossError = old.shorttermWinlossError;
    out.shorttermScoreError = old.shorttermScoreError;
    std::copy(old.policyProbs, old.policyProbs + NNPos::MAX_NN_POLICY_SIZE, out.policyProbs);
    out.policyOptimismUsed = old.policyOptimismUsed;
    out.nnXLen = old.nnXLen;
    out.nnYLen = old.nnYLen;
    leaf.cachedWithoutOwnerMap.reset();
  }
  else {
    policyToProbabilities(st, *leaf.board, *leaf.history, leaf.nextPlayer, leaf.nnInputParams, buf);
    valueToWhitePerspective(st, *leaf.board, *leaf.history, leaf.nextPlayer, out);
  }
  ownershipToWhitePerspective(st, *leaf.board, leaf.nextPlayer, out);
  out.nnHash = leaf.nnHash;
  if(st.cache != NULL)
    st.cache->set(buf.result);
  leaf.done = true;
}

// ---- the class ----------------------------------------------------------------------------------------------------------
NNEvaluator::NNEvaluator(
  const string& mName, const string& mFileName, const string& expectedSha256, Logger* lg, int maxBatchSz, int xLen, int yLen, bool rExactNNLen,
  bool iUseNHWC, int nnCacheSizePowerOfTwo, int nnMutexPoolSizePowerofTwo, bool skipNeuralNet, const string& homeDataDirOverride,
  enabled_t useFP16Mode, int numThr, const vector<int>& gpuIdxByServerThr, const string& rSeed, bool doRandomize, int defaultSymmetry,
  bool disableWarmup_, ConfigParser& cfg
)
  : modelName(mName), modelFileName(mFileName), nnXLen(xLen), nnYLen(yLen), requireExactNNLen(rExactNNLen),
    policySize(NNPos::getPolicySize(xLen, yLen)), inputsUseNHWC(iUseNHWC), usingFP16Mode(useFP16Mode), numThreads(numThr),
    gpuIdxByServerThread(gpuIdxByServerThr), randSeed(rSeed), debugSkipNeuralNet(skipNeuralNet), disableWarmup(disableWarmup_),
    computeContext(NULL), loadedModel(NULL), nnCacheTable(NULL), logger(lg), internalModelName(), modelVersion(-1), inputsVersion(-1),
    numInputMetaChannels(0), postProcessParams(), numServerThreadsEverSpawned(0), serverThreads(), maxBatchSize(maxBatchSz),
    m_numRowsProcessed(0), m_numBatchesProcessed(0), m_numCacheHits(0), bufferMutex(), isKilled(false), numServerThreadsStartingUp(0),
    mainThreadWaitingForSpawn(), numOngoingEvals(0), numWaitingEvals(0), numEvalsToAwaken(0), waitingForFinish(),
    currentDoRandomize(doRandomize), currentDefaultSymmetry(defaultSymmetry), currentBatchSize(maxBatchSz), queryQueue()
{
  if(nnXLen > NNPos::MAX_BOARD_LEN || nnYLen > NNPos::MAX_BOARD_LEN)
    throw StringError("Maximum supported nnEval board size is " + Global::intToString(NNPos::MAX_BOARD_LEN));
  if(maxBatchSize <= 0)
    throw StringError("maxBatchSize is negative: " + Global::intToString(maxBatchSize));
  if(gpuIdxByServerThread.size() != (size_t)numThreads)
    throw StringError("gpuIdxByServerThread.size() != numThreads");
  if(logger != NULL)
    logger->write("Initializing neural net buffer to be size " + Global::intToString(nnXLen) + " * " + Global::intToString(nnYLen) +
                  (requireExactNNLen ? " exactly" : " allowing smaller boards"));
  if(nnCacheSizePowerOfTwo >= 0)
    nnCacheTable = new NNCacheTable(nnCacheSizePowerOfTwo, nnMutexPoolSizePowerofTwo);

  if(debugSkipNeuralNet) {
    internalModelName = "random";
    modelVersion = NNModelVersion::defaultModelVersion;
    inputsVersion = NNModelVersion::getInputsVersion(modelVersion);
  }
  else {
    std::set<int> distinct(gpuIdxByServerThread.begin(), gpuIdxByServerThread.end());
    loadedModel = NeuralNet::loadModelFile(modelFileName, expectedSha256);
    const ModelDesc& desc = NeuralNet::getModelDesc(loadedModel);
    internalModelName = desc.name;
    modelVersion = desc.modelVersion;
    inputsVersion = NNModelVersion::getInputsVersion(modelVersion);
    numInputMetaChannels = desc.numInputMetaChannels;
    postProcessParams = desc.postProcessParams;
    computeContext = NeuralNet::createComputeContext(
      vector<int>(distinct.begin(), distinct.end()), logger, nnXLen, nnYLen, homeDataDirOverride, usingFP16Mode, loadedModel, cfg);
  }
  queryQueue.setReadOnly();  // never used: rows go to the leaf batcher from their owners' threads

  std::shared_ptr<EvalState> st = std::make_shared<EvalState>();
  st->nnXLen = nnXLen;
  st->nnYLen = nnYLen;
  st->policySize = policySize;
  st->modelVersion = modelVersion;
  st->inputsVersion = inputsVersion;
  st->numInputMetaChannels = numInputMetaChannels;
  st->requireExactNNLen = requireExactNNLen;
  st->debugSkipNeuralNet = debugSkipNeuralNet;
  st->post = postProcessParams;
  st->cache = nnCacheTable;
  st->logger = logger;
  st->modelName = modelName;
  st->modelFileName = modelFileName;
  st->cacheHits = &m_numCacheHits;
  st->doRandomize = &currentDoRandomize;
  st->defaultSymmetry = &currentDefaultSymmetry;
  std::unique_lock<std::shared_mutex> lock(registryMutex);
  registry[this] = st;
}

NNEvaluator::~NNEvaluator() {
  killServerThreads();
  {
    std::unique_lock<std::shared_mutex> lock(registryMutex);
    registry.erase(this);
  }
  if(computeContext != NULL)
    NeuralNet::freeComputeContext(computeContext);
  if(loadedModel != NULL)
    NeuralNet::freeLoadedModel(loadedModel);
  delete nnCacheTable;
}

// "Server threads" of this evaluator are the leaf ports: one persistent batcher per distinct device, with as many device
// batches in flight as the configuration asked server threads for on that device (+1, at least 2). The seed string of
// port k is the reference's for its server thread k, so that a single-threaded caller sees the same random symmetries.
void NNEvaluator::spawnServerThreads() {
  const std::shared_ptr<EvalState> st = stateOf(this);
  if(!st->ports.empty() || st->nnlessRand != nullptr)
    throw StringError("NNEvaluator::spawnServerThreads called when threads were already running!");
  {
    std::lock_guard<std::mutex> lock(st->flightMutex);
    st->stopping = false;
  }
  std::vector<int> distinct;
  std::vector<int> threadsOn;
  for(int i = 0; i < numThreads; i++) {
    const int gpu = gpuIdxByServerThread[i];
    size_t k = 0;
    while(k < distinct.size() && distinct[k] != gpu) k++;
    if(k == distinct.size()) {
      distinct.push_back(gpu);
      threadsOn.push_back(0);
    }
    threadsOn[k]++;
  }
  std::lock_guard<std::mutex> lock(bufferMutex);
  serverThreadsIsUsingFP16.assign(numThreads, 0);
  if(debugSkipNeuralNet) {
    st->nnlessRand.reset(new Rand(randSeed + ":NNEvalServerThread:" + Global::intToString(numServerThreadsEverSpawned)));
    numServerThreadsEverSpawned += numThreads;
    return;
  }
  for(size_t k = 0; k < distinct.size(); k++) {
    std::unique_ptr<PortSlot> slot(new PortSlot());
    slot->gpuIdx = distinct[k];
    slot->rand.reset(new Rand(randSeed + ":NNEvalServerThread:" + Global::intToString(numServerThreadsEverSpawned + (int)k)));
    slot->port = KatamxLeaf::openPort(computeContext, loadedModel, logger, maxBatchSize, std::min(8, std::max(2, threadsOn[k] + 1)), distinct[k]);
    slot->usingFP16 = KatamxLeaf::isUsingFP16(slot->port);
    st->ports.push_back(std::move(slot));
  }
  numServerThreadsEverSpawned += numThreads;
  {
    const char* pol = getenv("KATAMX_PORT_POLICY");
    if(pol != NULL && string(pol) != "spread" && string(pol) != "fill" && string(pol) != "")
      throw StringError("KATAMX_PORT_POLICY must be spread or fill, not " + string(pol));
    st->fillFirst = pol != NULL && string(pol) == "fill";
    st->fillTarget = std::max(1, std::min(maxBatchSize, 256));
    if(const char* e = getenv("KATAMX_PORT_FILL_ROWS"))
      st->fillTarget = std::max(1, atoi(e));
    if(logger != NULL && st->ports.size() > 1)
      logger->write(
        "katamx leaf ports: " + Global::intToString((int)st->ports.size()) + " devices, rows go to " +
        (st->fillFirst ? "the first port below " + Global::intToString(st->fillTarget) + " rows in flight (KATAMX_PORT_POLICY=fill)"
                       : string("the port with the fewest rows in flight (KATAMX_PORT_POLICY=spread)")));
  }
  for(int i = 0; i < numThreads; i++)
    for(const auto& slot : st->ports)
      if(slot->gpuIdx == gpuIdxByServerThread[i]) serverThreadsIsUsingFP16[i] = slot->usingFP16 ? 1 : 0;
}

void NNEvaluator::killServerThreads() {
  std::shared_ptr<EvalState> st;
  {
    std::shared_lock<std::shared_mutex> lock(registryMutex);
    auto it = registry.find(this);
    if(it == registry.end())
      return;
    st = it->second;
  }
  {
    std::lock_guard<std::mutex> lock(st->flightMutex);
    st->stopping = true;
  }
  st->flightChanged.notify_all();
  // rows and batches of the closing ports stay in the evaluator's counters
  for(auto& slot : st->ports) {
    uint64_t rows = 0, batches = 0;
    KatamxLeaf::stats(slot->port, rows, batches);
    m_numRowsProcessed.fetch_add(rows, std::memory_order_relaxed);
    m_numBatchesProcessed.fetch_add(batches, std::memory_order_relaxed);
    KatamxLeaf::closePort(slot->port);
  }
  st->ports.clear();
  if(st->nnlessRand != nullptr) {
    const uint64_t rows = st->nnlessRows.exchange(0);
    m_numRowsProcessed.fetch_add(rows, std::memory_order_relaxed);
    m_numBatchesProcessed.fetch_add(rows, std::memory_order_relaxed);
    st->nnlessRand.reset();
  }
  {
    std::lock_guard<std::mutex> lock(bufferMutex);
    serverThreadsIsUsingFP16.clear();
  }
  {
    std::lock_guard<std::mutex> lock(st->flightMutex);
    testAssert(st->inFlight == 0);
    st->stopping = false;
  }
}

void NNEvaluator::setNumThreads(const vector<int>& gpuIdxByServerThr) {
  if(!stateOf(this)->ports.empty())
    throw StringError("NNEvaluator::setNumThreads called when threads were already running!");
  numThreads = (int)gpuIdxByServerThr.size();
  gpuIdxByServerThread = gpuIdxByServerThr;
}

void NNEvaluator::serve(NNServerBuf&, Rand&, int, int) {
  throw StringError("katamx NNEvaluator has no server threads: rows are submitted to the leaf batcher by the threads that own them");
}
void NNEvaluator::maybeWarmupComputeHandle(ComputeHandle*, int) {}

void NNEvaluator::fillRowBufs(
  const Board& board, const BoardHistory& history, Player nextPlayer, const SGFMetadata* sgfMeta, const MiscNNInputParams& nnInputParams,
  NNResultBuf& buf
) const {
  // the declared contract of this method is the fp32 row in buf.rowSpatialBuf: expand this repository's bit planes into it
  const std::shared_ptr<EvalState> sp = stateOf(this);
  EvalState& st = *sp;
  uint8_t packedRow[KatamxFeatures::MAX_PACKED_ROW_BYTES];
  if(featurise(st, board, history, nextPlayer, sgfMeta, nnInputParams, buf, packedRow)) {
    const int numPlanes = NNModelVersion::getNumSpatialFeatures(st.modelVersion);
    buf.rowSpatialBuf.resize((size_t)numPlanes * st.nnXLen * st.nnYLen);
    KatamxFeatures::unpackToNHWC(packedRow, st.nnXLen, st.nnYLen, numPlanes, buf.rowSpatialBuf.data());
  }
}

void NNEvaluator::evaluate(
  const Board& board, const BoardHistory& history, Player nextPlayer, const MiscNNInputParams& nnInputParams, NNResultBuf& buf, bool skipCache,
  bool includeOwnerMap
) {
  evaluate(board, history, nextPlayer, NULL, nnInputParams, buf, skipCache, includeOwnerMap);
}
void NNEvaluator::evaluate(
  const Board& board, const BoardHistory& history, Player nextPlayer, const SGFMetadata* sgfMeta, const MiscNNInputParams& nnInputParams,
  NNResultBuf& buf, bool skipCache, bool includeOwnerMap
) {
  KatamxNNEval::Leaf leaf;
  KatamxNNEval::begin(*this, board, history, nextPlayer, sgfMeta, nnInputParams, buf, skipCache, includeOwnerMap, leaf);
  // A search thread that runs on a fiber (katamx_fibers.h) gives its OS thread to the next descent while the row is on the device
  if(leaf.inFlight && leaf.port != NULL && KatamxFibers::onFiber()) {
    try {
      leaf.ticketCollected = KatamxFibers::park(leaf.port, leaf.ticket);
    }
    catch(...) {
      leaf.ticketCollected = true;
      leaf.collectError = std::current_exception();
    }
  }
  KatamxNNEval::finish(*this, leaf);
}

// The sampled symmetries are all in flight at once (they land in the same device batch) instead of one after the other.
std::shared_ptr<NNOutput>* NNEvaluator::averageMultipleSymmetries(
  const Board& board, const BoardHistory& history, Player nextPlayer, const SGFMetadata* sgfMeta, const MiscNNInputParams& baseNNInputParams,
  NNResultBuf& buf, bool includeOwnerMap, Rand& rand, int numSymmetriesToSample
) {
  std::array<int, SymmetryHelpers::NUM_SYMMETRIES> order;
  std::iota(order.begin(), order.end(), 0);
  std::vector<std::unique_ptr<NNResultBuf>> bufs;
  std::vector<std::unique_ptr<KatamxNNEval::Leaf>> leaves;
  std::exception_ptr failure;
  for(int i = 0; i < numSymmetriesToSample; i++) {
    std::swap(order[i], order[rand.nextInt(i, SymmetryHelpers::NUM_SYMMETRIES - 1)]);
    MiscNNInputParams params = baseNNInputParams;
    params.symmetry = order[i];
    bufs.emplace_back(new NNResultBuf());
    leaves.emplace_back(new KatamxNNEval::Leaf());
    // no cache: nothing says which symmetry a cached entry was computed with
    try {
      KatamxNNEval::begin(*this, board, history, nextPlayer, sgfMeta, params, *bufs.back(), true, includeOwnerMap, *leaves.back());
    }
    catch(...) {
      // (the batcher is shutting down, a submit failed) The leaves begun so far are on the device, and the batcher's completion
      // thread will write into their Leaf / NNOutput buffers: they must be collected before those buffers go out of scope
      failure = std::current_exception();
      leaves.pop_back();
      bufs.pop_back();
      break;
    }
  }
  vector<std::shared_ptr<NNOutput>> results;
  for(size_t i = 0; i < leaves.size(); i++) {
    try {
      KatamxNNEval::finish(*this, *leaves[i]);
      results.push_back(std::move(bufs[i]->result));
    }
    catch(...) {
      if(!failure) failure = std::current_exception();  // the remaining tickets are still collected
    }
  }
  if(failure)
    std::rethrow_exception(failure);
  buf.hasResult = false;
  return new std::shared_ptr<NNOutput>(new NNOutput(results));
}

void NNEvaluator::waitForNextNNEvalIfAny() {
  // on a fiber the evaluations "in flight" may all belong to parked fibers of this very OS thread: let them finish instead of
  // sleeping on completions nobody else would produce
  if(KatamxFibers::yieldToOthers())
    return;
  const std::shared_ptr<EvalState> st = stateOf(this);
  std::unique_lock<std::mutex> lock(st->flightMutex);
  if(st->inFlight <= 0)
    return;
  const uint64_t seen = st->completions;
  st->flightChanged.wait(lock, [&] { return st->completions != seen || st->stopping; });
}

// ---- accessors ----------------------------------------------------------------------------------------------------------
string NNEvaluator::getModelName() const { return modelName; }
string NNEvaluator::getModelFileName() const { return modelFileName; }
string NNEvaluator::getInternalModelName() const { return internalModelName; }
Logger* NNEvaluator::getLogger() { return logger; }
bool NNEvaluator::isNeuralNetLess() const { return debugSkipNeuralNet; }
int NNEvaluator::getMaxBatchSize() const { return maxBatchSize; }
int NNEvaluator::getCurrentBatchSize() const { return currentBatchSize.load(std::memory_order_acquire); }
void NNEvaluator::setCurrentBatchSize(int batchSize) {
  if(batchSize <= 0 || batchSize > maxBatchSize)
    throw StringError("Invalid setting for batch size");
  currentBatchSize.store(batchSize, std::memory_order_release);  // informational: the batcher seals greedily whatever is waiting
}
bool NNEvaluator::requiresSGFMetadata() const { return numInputMetaChannels > 0; }
int NNEvaluator::getNumGpus() const { return (int)getGpuIdxs().size(); }
int NNEvaluator::getNumServerThreads() const { return (int)gpuIdxByServerThread.size(); }
std::set<int> NNEvaluator::getGpuIdxs() const { return std::set<int>(gpuIdxByServerThread.begin(), gpuIdxByServerThread.end()); }
int NNEvaluator::getNNXLen() const { return nnXLen; }
int NNEvaluator::getNNYLen() const { return nnYLen; }
bool NNEvaluator::getRequireExactNNLen() const { return requireExactNNLen; }
int NNEvaluator::getModelVersion() const { return modelVersion; }
double NNEvaluator::getTrunkSpatialConvDepth() const { return NeuralNet::getModelDesc(loadedModel).getTrunkSpatialConvDepth(); }
enabled_t NNEvaluator::getUsingFP16Mode() const { return usingFP16Mode; }
bool NNEvaluator::supportsShorttermError() const { return modelVersion >= 9; }
bool NNEvaluator::modelPreferPassAliveUnderSuicideRules() const {
  return loadedModel != NULL && NeuralNet::getModelDesc(loadedModel).preferPassAliveUnderSuicideRules;
}
bool NNEvaluator::getDoRandomize() const { return currentDoRandomize.load(std::memory_order_acquire); }
int NNEvaluator::getDefaultSymmetry() const { return currentDefaultSymmetry.load(std::memory_order_acquire); }
void NNEvaluator::setDoRandomize(bool b) { currentDoRandomize.store(b, std::memory_order_release); }
void NNEvaluator::setDefaultSymmetry(int s) { currentDefaultSymmetry.store(s, std::memory_order_release); }
Rules NNEvaluator::getSupportedRules(const Rules& desiredRules, bool& supported) const {
  if(loadedModel == NULL) {
    supported = true;
    return desiredRules;
  }
  return NeuralNet::getModelDesc(loadedModel).getSupportedRules(desiredRules, supported);
}
bool NNEvaluator::isAnyThreadUsingFP16() const {
  std::lock_guard<std::mutex> lock(bufferMutex);
  for(int v : serverThreadsIsUsingFP16)
    if(v) return true;
  return false;
}

// counters = what closed ports left behind + what the open ones report now
uint64_t NNEvaluator::numRowsProcessed() const {
  uint64_t total = m_numRowsProcessed.load(std::memory_order_relaxed);
  const std::shared_ptr<EvalState> st = stateOf(this);
  for(const auto& slot : st->ports) {
    uint64_t rows = 0, batches = 0;
    KatamxLeaf::stats(slot->port, rows, batches);
    total += rows;
  }
  return total + st->nnlessRows.load(std::memory_order_relaxed);
}
uint64_t NNEvaluator::numBatchesProcessed() const {
  uint64_t total = m_numBatchesProcessed.load(std::memory_order_relaxed);
  const std::shared_ptr<EvalState> st = stateOf(this);
  for(const auto& slot : st->ports) {
    uint64_t rows = 0, batches = 0;
    KatamxLeaf::stats(slot->port, rows, batches);
    total += batches;
  }
  return total + st->nnlessRows.load(std::memory_order_relaxed);
}
double NNEvaluator::averageProcessedBatchSize() const { return (double)numRowsProcessed() / (double)numBatchesProcessed(); }
uint64_t NNEvaluator::numCacheHits() const { return m_numCacheHits.load(std::memory_order_relaxed); }
void NNEvaluator::clearStats() {
  // the ports' own counters cannot be reset: remember them as a negative offset
  uint64_t rows = 0, batches = 0;
  const std::shared_ptr<EvalState> st = stateOf(this);
  for(const auto& slot : st->ports) {
    uint64_t r = 0, b = 0;
    KatamxLeaf::stats(slot->port, r, b);
    rows += r;
    batches += b;
  }
  const uint64_t nnless = st->nnlessRows.load(std::memory_order_relaxed);
  m_numRowsProcessed.store(0 - rows - nnless);
  m_numBatchesProcessed.store(0 - batches - nnless);
  m_numCacheHits.store(0);
}
void NNEvaluator::clearCache() {
  if(nnCacheTable != NULL)
    nnCacheTable->clear();
}

// "kata1-b18c384nbt-s9131461376-d4087399203" -> "b18c384nbt-s9131M": drop the run name and the data count, shorten the steps
string NNEvaluator::getAbbrevInternalModelName() const {
  auto shortenCount = [](const string& piece, string& out) {
    // one leading letter, then digits only
    if(piece.size() < 2 || Global::isDigit(piece[0]))
      return false;
    int64_t n;
    if(!Global::tryStringToInt64(piece.substr(1), n))
      return false;
    const string prefix = piece.substr(0, 1);
    out = piece;
    if(n >= 10000) out = prefix + std::to_string(n / 1000) + "K";
    if(n >= 10000000) out = prefix + std::to_string(n / 1000000) + "M";
    return true;
  };
  std::vector<string> kept;
  for(const string& piece : Global::split(getInternalModelName(), '-')) {
    string shortened;
    if(piece == "kata1") continue;
    if(piece.size() > 1 && piece[0] == 'd' && shortenCount(piece, shortened)) continue;
    if(piece.size() > 1 && piece[0] == 's' && shortenCount(piece, shortened)) kept.push_back(shortened);
    else kept.push_back(piece);
  }
  return Global::concat(kept, "-");
}
