// katamx_features.cpp — inputs version 7 written as bit planes (see katamx_features.h).
//
// What each plane and global MEANS is fixed by the reference's featuriser, because trained nets read it:
// NNInputs::fillRowV7, cpp/neuralnet/nninputs.cpp:2288-2731 (ladders: iterLadders, :815-866). How it is produced here differs:
// the destination is 22 bit planes, not 7 942 floats; the per-cell planes (board, stones, liberties, ko bans, second-encore
// stones) are gathered in ONE sweep over the board; move history is a loop over plies; the ladder scan remembers its verdict
// per chain in an array indexed by the chain's head (the reference searches a list per stone), is not repeated when two of
// the three boards it is asked about are the same object, and is looked up instead of repeated for the two older boards when
// their positions were featurised before (LadderMemo); the scoring variants are a small table.
#include "katamx_features.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "game/rules.h"

namespace {

struct BitPlanes {
  uint8_t* base;
  int planeBytes;
  inline void set(int plane, int pos) const { base[plane * planeBytes + (pos >> 3)] |= (uint8_t)(0x80u >> (pos & 7)); }
  inline uint8_t* plane(int p) const { return base + p * planeBytes; }
};

// Plane numbers of inputs version 7 (nninputs.cpp:2321-2592).
enum : int {
  PL_ON_BOARD = 0, PL_OWN = 1, PL_OPP = 2, PL_LIBS1 = 3, PL_LIBS2 = 4, PL_LIBS3 = 5,
  PL_KO_BAN = 6, PL_ENCORE_KO_RECAP = 7,  // plane 8 is never written by the reference either
  PL_HISTORY1 = 9,                         // 9..13: the last five moves
  PL_LADDERED = 14, PL_LADDERED_PREV = 15, PL_LADDERED_PREV2 = 16, PL_LADDER_WORKS = 17,
  PL_AREA_OWN = 18, PL_AREA_OPP = 19, PL_ENCORE2_OWN = 20, PL_ENCORE2_OPP = 21,
};
enum : int {
  GL_PASS_HISTORY1 = 0,  // 0..4
  GL_KOMI = 5, GL_KO_RULE_A = 6, GL_KO_RULE_B = 7, GL_SUICIDE = 8, GL_TERRITORY = 9, GL_TAX_A = 10, GL_TAX_B = 11,
  GL_ENCORE1 = 12, GL_ENCORE2 = 13, GL_PASS_ENDS_PHASE = 14, GL_PDA_ON = 15, GL_PDA = 16, GL_BUTTON = 17, GL_KOMI_PARITY_WAVE = 18,
};

// Chains in or one move from an inescapable atari (nninputs.cpp:815-866). Every chain with one or two liberties is read
// once, in the order its first stone is met going row by row, on ONE scratch copy of the board - the reference's order, kept
// because the searches share that copy. marks: the stones of the laddered chains. works (may be NULL): for laddered chains
// of `worksFor` with two liberties, the attacker's moves that start the ladder (nninputs.cpp:2547-2556).
void markLadders(const Board& board, int nnXLen, int nnYLen, const BitPlanes& out, int markPlane, int worksPlane, Player worksFor) {
  const int xSize = board.x_size, ySize = board.y_size;
  enum : uint8_t { UNREAD = 0, ESCAPES = 1, CAUGHT = 2 };
  uint8_t verdict[Board::MAX_ARR_SIZE];
  memset(verdict, UNREAD, sizeof(verdict));
  Board scratch(board);
  std::vector<Loc> searchBuf, works;
  for(int y = 0; y < ySize; y++) {
    for(int x = 0; x < xSize; x++) {
      const Loc loc = Location::getLoc(x, y, xSize);
      const Color c = board.colors[loc];
      if(c != P_BLACK && c != P_WHITE)
        continue;
      const int libs = board.getNumLiberties(loc);
      if(libs != 1 && libs != 2)
        continue;
      uint8_t& v = verdict[board.chain_head[loc]];
      if(v == UNREAD) {
        bool caught;
        if(libs == 1)
          caught = scratch.searchIsLadderCaptured(loc, true, searchBuf);
        else {
          works.clear();
          caught = scratch.searchIsLadderCapturedAttackerFirst2Libs(loc, searchBuf, works);
          if(caught && worksPlane >= 0 && c == worksFor)
            for(const Loc w : works) out.set(worksPlane, NNPos::locToPos(w, xSize, nnXLen, nnYLen));
        }
        v = caught ? CAUGHT : ESCAPES;
      }
      if(v == CAUGHT)
        out.set(markPlane, NNPos::xyToPos(x, y, nnXLen));
    }
  }
}

// The ladder plane of a position - the stones of its caught chains - is a function of the stones, the simple-ko point (the
// reading respects it) and the buffer geometry, nothing else. A process-wide direct-mapped table keeps the planes of recently
// featurised positions under that key (128-bit Zobrist hash, as the reference's NN cache trusts it, cpp/neuralnet/nneval.cpp:
// 1273-1353); striped mutexes, one cache line per entry at 19x19. KATAMX_LADDER_MEMO_LOG2 sets the size (default 2^16 entries =
// 4 MB; 0 = off).
class LadderMemo {
 public:
  static LadderMemo& instance() {
    static LadderMemo memo;
    return memo;
  }
  bool lookup(const Board& board, int nnXLen, int nnYLen, uint8_t* plane, int planeBytes) {
    if(entries_.empty())
      return false;
    const Hash128 key = keyOf(board, nnXLen, nnYLen);
    Entry& e = entries_[key.hash0 & mask_];
    std::lock_guard<std::mutex> lock(stripes_[key.hash0 & (NUM_STRIPES - 1)]);
    if(!e.valid || e.key != key)
      return false;
    memcpy(plane, e.bits, planeBytes);
    return true;
  }
  void store(const Board& board, int nnXLen, int nnYLen, const uint8_t* plane, int planeBytes) {
    if(entries_.empty())
      return;
    const Hash128 key = keyOf(board, nnXLen, nnYLen);
    Entry& e = entries_[key.hash0 & mask_];
    std::lock_guard<std::mutex> lock(stripes_[key.hash0 & (NUM_STRIPES - 1)]);
    e.key = key;
    e.valid = 1;
    memcpy(e.bits, plane, planeBytes);
  }

 private:
  static constexpr int MAX_PLANE_BYTES = (NNPos::MAX_BOARD_AREA + 7) / 8;
  static constexpr int NUM_STRIPES = 1024;
  struct alignas(64) Entry {
    Hash128 key;
    uint8_t bits[MAX_PLANE_BYTES];
    uint8_t valid = 0;
  };
  std::vector<Entry> entries_;
  uint64_t mask_ = 0;
  std::mutex stripes_[NUM_STRIPES];

  LadderMemo() {
    int log2 = 16;
    if(const char* env = getenv("KATAMX_LADDER_MEMO_LOG2"))
      log2 = atoi(env);
    if(log2 > 0) {
      log2 = std::min(std::max(log2, 10), 24);
      entries_.resize((size_t)1 << log2);
      mask_ = ((uint64_t)1 << log2) - 1;
    }
  }
  static Hash128 keyOf(const Board& board, int nnXLen, int nnYLen) {
    Hash128 key = board.pos_hash ^ Board::ZOBRIST_KO_LOC_HASH[board.ko_loc];  // pos_hash covers the stones and the board's size
    key.hash1 ^= (uint64_t)(nnXLen * 64 + nnYLen) * 0x9E3779B97F4A7C15ULL;
    return key;
  }
};

// Which "who owns what if the game stopped now" map planes 18/19 show, by scoring and tax rule (nninputs.cpp:2371-2428).
struct AreaMode {
  bool shown;             // the planes exist under these rules at this stage
  bool independentLife;   // Board::calculateIndependentLifeArea instead of Board::calculateArea
  bool keepTerritories, keepStones;
};
AreaMode areaModeFor(const Rules& rules, int encorePhase) {
  const bool taxed = rules.taxRule == Rules::TAX_SEKI || rules.taxRule == Rules::TAX_ALL;
  if(rules.taxRule != Rules::TAX_NONE && !taxed)
    ASSERT_UNREACHABLE;
  if(rules.scoringRule == Rules::SCORING_AREA)
    return taxed ? AreaMode{true, true, false, true} : AreaMode{true, false, false, false};
  if(rules.scoringRule == Rules::SCORING_TERRITORY)  // nothing to show until the stage where scoring matters
    return AreaMode{encorePhase >= 2, true, !taxed, false};
  ASSERT_UNREACHABLE;
  return AreaMode{false, false, false, false};
}

}  // namespace

void KatamxFeatures::fillPackedV7(
  const Board& board, const BoardHistory& hist, Player nextPlayer, const MiscNNInputParams& params,
  int nnXLen, int nnYLen, uint8_t* packed, float* rowGlobal
) {
  assert(nnXLen <= NNPos::MAX_BOARD_LEN && nnYLen <= NNPos::MAX_BOARD_LEN);
  assert(board.x_size <= nnXLen && board.y_size <= nnYLen);
  const BitPlanes out{packed, planeBytes(nnXLen, nnYLen)};
  memset(packed, 0, packedRowBytes(nnXLen, nnYLen));
  std::fill(rowGlobal, rowGlobal + NUM_GLOBALS_V7, 0.0f);

  const Player pla = nextPlayer, opp = getOpp(pla);
  const int xSize = board.x_size, ySize = board.y_size;
  const Rules& rules = hist.rules;
  const bool inEncore = hist.encorePhase > 0, inSecondEncore = hist.encorePhase >= 2;

  // ---- the planes that are a function of one cell: 0-5 board, stones and liberties; 6-7 ko bans; 20-21 second-encore stones
  for(int y = 0; y < ySize; y++) {
    for(int x = 0; x < xSize; x++) {
      const Loc loc = Location::getLoc(x, y, xSize);
      const int pos = NNPos::xyToPos(x, y, nnXLen);
      out.set(PL_ON_BOARD, pos);
      const Color stone = board.colors[loc];
      if(stone == pla || stone == opp) {
        out.set(stone == pla ? PL_OWN : PL_OPP, pos);
        const int libs = board.getNumLiberties(loc);
        if(libs >= 1 && libs <= 3)
          out.set(PL_LIBS1 + libs - 1, pos);
      }
      // before the encore plane 6 is "may not play here because of ko" (simple ko point or superko), in the encore it is the
      // superko ban alone and plane 7 the no-second-recapture marks
      if(hist.superKoBanned[loc] || (!inEncore && loc == board.ko_loc))
        out.set(PL_KO_BAN, pos);
      if(inEncore && hist.koRecapBlocked[loc])
        out.set(PL_ENCORE_KO_RECAP, pos);
      if(inSecondEncore) {
        const Color then = hist.secondEncoreStartColors[loc];
        if(then == pla || then == opp)
          out.set(then == pla ? PL_ENCORE2_OWN : PL_ENCORE2_OPP, pos);
      }
    }
  }

  // ---- planes 18-19: area as it stands, and with it the score if the game were counted now
  bool countedNowWouldNotWin = false;
  const AreaMode mode = areaModeFor(rules, hist.encorePhase);
  if(mode.shown) {
    Color area[Board::MAX_ARR_SIZE];
    int plaPoints = 0;
    const bool suicideLegal = params.getSuicideLegalForPassAlive(hist);
    if(!mode.independentLife)
      board.calculateArea(area, true, true, true, suicideLegal);
    else {
      int whiteMinusBlackRegions = 0;
      board.calculateIndependentLifeArea(area, whiteMinusBlackRegions, mode.keepTerritories, mode.keepStones, suicideLegal);
      if(rules.taxRule == Rules::TAX_ALL)  // two points of group tax per independently living region
        plaPoints = (pla == P_WHITE ? -2 : 2) * whiteMinusBlackRegions;
    }
    const bool territoryScoring = rules.scoringRule == Rules::SCORING_TERRITORY;
    for(int y = 0; y < ySize; y++) {
      for(int x = 0; x < xSize; x++) {
        const Loc loc = Location::getLoc(x, y, xSize);
        Color owner = area[loc];
        // territory scoring (second encore by now): a stone that stood at the start of this encore and still stands counts too
        if(owner != pla && owner != opp && territoryScoring && board.colors[loc] == hist.secondEncoreStartColors[loc])
          owner = board.colors[loc];
        if(owner == pla || owner == opp) {
          out.set(owner == pla ? PL_AREA_OWN : PL_AREA_OPP, NNPos::xyToPos(x, y, nnXLen));
          plaPoints += owner == pla ? 1 : -1;
        }
      }
    }
    const float scoreNow = (float)plaPoints + hist.currentSelfKomi(pla, params.drawEquivalentWinsForWhite);
    countedNowWouldNotWin = scoreNow <= 0.0;
  }

  // ---- how much move history the net is shown (nninputs.cpp:2467-2491)
  int historyCap = 5;
  bool hidePassEndsPhase = false;
  if(hist.passWouldEndGame(board, nextPlayer) &&
     (params.conservativePassAndIsRoot ||                             // the root pretends that passing does not end the game
      hist.shouldSuppressEndGameFromFriendlyPass(board, nextPlayer) ||  // friendly-pass settings, deeper in the tree
      (params.enablePassingHacks && countedNowWouldNotWin))) {          // do not let a losing net pass the game away
    historyCap = 0;
    hidePassEndsPhase = true;
  }
  else if(hist.isGameFinished || hist.isPastNormalPhaseEnd)
    historyCap = 1;  // one of the closing passes
  historyCap = std::min(historyCap, params.maxHistory);

  // ---- planes 9-13 / globals 0-4: the last plies, alternating players, never across a phase change
  int pliesShown = 0;
  {
    const std::vector<Move>& moves = hist.moveHistory;
    const int numMoves = (int)moves.size();
    assert(numMoves >= hist.numApproxValidTurnsThisPhase);
    const int usable = std::min(std::min(historyCap, hist.numApproxValidTurnsThisPhase), numMoves);
    for(int ply = 1; ply <= usable; ply++) {
      const Move& m = moves[numMoves - ply];
      if(m.pla != (ply % 2 == 1 ? opp : pla))
        break;
      pliesShown = ply;
      if(m.loc == Board::PASS_LOC)
        rowGlobal[GL_PASS_HISTORY1 + ply - 1] = 1.0f;
      else if(m.loc != Board::NULL_LOC)
        out.set(PL_HISTORY1 + ply - 1, NNPos::locToPos(m.loc, xSize, nnXLen, nnYLen));
    }
  }

  // ---- planes 14-17: ladders now, one and two plies ago (boards older than the history shown are replaced by the newer one).
  // The two older boards are, in a search, the parent's and the grandparent's positions, whose ladders were read when THEY were
  // featurised: the memo hands their planes back (ladder reading is 3/4 of a row's cost, and two of its three scans are these).
  LadderMemo& memo = LadderMemo::instance();
  markLadders(board, nnXLen, nnYLen, out, PL_LADDERED, PL_LADDER_WORKS, opp);
  memo.store(board, nnXLen, nnYLen, out.plane(PL_LADDERED), out.planeBytes);
  const Board* newer = &board;
  int newerPlane = PL_LADDERED;
  for(int ply = 1; ply <= 2; ply++) {
    const int plane = ply == 1 ? PL_LADDERED_PREV : PL_LADDERED_PREV2;
    const Board* older = pliesShown < ply ? newer : &hist.getRecentBoard(ply);
    if(older == newer)
      memcpy(out.plane(plane), out.plane(newerPlane), out.planeBytes);
    else if(!memo.lookup(*older, nnXLen, nnYLen, out.plane(plane), out.planeBytes)) {
      markLadders(*older, nnXLen, nnYLen, out, plane, -1, C_EMPTY);
      memo.store(*older, nnXLen, nnYLen, out.plane(plane), out.planeBytes);
    }
    newer = older;
    newerPlane = plane;
  }

  // ---- globals 5-18 (nninputs.cpp:2595-2730)
  float selfKomi = hist.currentSelfKomi(nextPlayer, params.drawEquivalentWinsForWhite);
  {
    const float komiBound = (float)(xSize * ySize) + NNPos::KOMI_CLIP_RADIUS;
    selfKomi = std::min(std::max(selfKomi, -komiBound), komiBound);
  }
  rowGlobal[GL_KOMI] = selfKomi / 20.0f;

  switch(rules.koRule) {
    case Rules::KO_SIMPLE: break;
    case Rules::KO_POSITIONAL:
    case Rules::KO_SPIGHT: rowGlobal[GL_KO_RULE_A] = 1.0f; rowGlobal[GL_KO_RULE_B] = 0.5f; break;
    case Rules::KO_SITUATIONAL: rowGlobal[GL_KO_RULE_A] = 1.0f; rowGlobal[GL_KO_RULE_B] = -0.5f; break;
    default: ASSERT_UNREACHABLE;
  }
  if(rules.multiStoneSuicideLegal)
    rowGlobal[GL_SUICIDE] = 1.0f;
  if(rules.scoringRule == Rules::SCORING_TERRITORY)
    rowGlobal[GL_TERRITORY] = 1.0f;
  if(rules.taxRule != Rules::TAX_NONE)
    rowGlobal[GL_TAX_A] = 1.0f;
  if(rules.taxRule == Rules::TAX_ALL)
    rowGlobal[GL_TAX_B] = 1.0f;
  if(inEncore)
    rowGlobal[GL_ENCORE1] = 1.0f;
  if(inSecondEncore)
    rowGlobal[GL_ENCORE2] = 1.0f;
  if(!hidePassEndsPhase && hist.passWouldEndPhase(board, nextPlayer))
    rowGlobal[GL_PASS_ENDS_PHASE] = 1.0f;
  if(params.playoutDoublingAdvantage != 0) {  // its own flag: training treats "exactly zero" as a different regime
    rowGlobal[GL_PDA_ON] = 1.0f;
    rowGlobal[GL_PDA] = (float)(0.5 * params.playoutDoublingAdvantage);
  }
  if(hist.hasButton)
    rowGlobal[GL_BUTTON] = 1.0f;

  // Komi parity: a triangle wave of period 2 in the komi seen by the player to move, rising through the komi values at which
  // a draw is possible (those have the parity of the board area), so that "half a point more" is linear for the net.
  if(rules.scoringRule == Rules::SCORING_AREA || inSecondEncore) {
    // drawable komis are even on even-area boards, odd otherwise; `below` = the nearest drawable komi not above selfKomi
    const bool evenArea = (xSize * ySize) % 2 == 0;
    const float below = evenArea ? std::floor(selfKomi / 2.0f) * 2.0f : std::floor((selfKomi - 1.0f) / 2.0f) * 2.0f + 1.0f;
    float delta = selfKomi - below;
    assert(delta >= -0.0001f && delta <= 2.0001f);
    if(delta < 0.0f) delta = 0.0f;
    if(delta > 2.0f) delta = 2.0f;
    rowGlobal[GL_KOMI_PARITY_WAVE] = delta < 0.5f ? delta : delta < 1.5f ? 1.0f - delta : delta - 2.0f;
  }
}

void KatamxFeatures::unpackToNHWC(const uint8_t* packed, int nnXLen, int nnYLen, int numPlanes, float* rowNHWC) {
  const int cells = nnXLen * nnYLen, pb = planeBytes(nnXLen, nnYLen);
  for(int pos = 0; pos < cells; pos++) {
    const int byte = pos >> 3, shift = 7 - (pos & 7);
    float* cell = rowNHWC + (size_t)pos * numPlanes;
    for(int p = 0; p < numPlanes; p++)
      cell[p] = (float)((packed[p * pb + byte] >> shift) & 1);
  }
}
