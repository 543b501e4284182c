// features_selftest.cpp — test infrastructure: KatamxFeatures::fillPackedV7 (integration/katamx_features.cpp) against the
// reference's NNInputs::fillRowV7 (cpp/neuralnet/nninputs.cpp:2288-2731), both linked into this one binary, over random games.
//
//   features_selftest <games> <seed> [time | threads N]
//
// Every game draws a board size (square and rectangular, smaller than or equal to the net's buffer), a full rule set (ko rule x
// scoring x tax x suicide x button x handicap bonus x friendly pass, komi incl. extreme and integer values), optional handicap
// stones, and is played with random legal moves and a pass rate that rises late, so that territory-scoring games walk through
// both encore phases and games end by passes. At every position both featurisers run under several MiscNNInputParams (root
// conservative pass, passing hacks, history limits, playout doubling advantage, draw equivalence, pass-alive override) and for
// both colours to move; the bit planes are expanded (unpackToNHWC) and must equal the reference's fp32 row byte for byte, the
// globals likewise. Exit code 0 = no difference. With `time`, both are also timed on the collected positions.
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "core/global.h"
#include "core/rand.h"
#include "game/board.h"
#include "game/boardhistory.h"
#include "game/rules.h"
#include "neuralnet/nninputs.h"

#include "katamx_features.h"

namespace {

struct Sizes { int x, y; };
const Sizes BOARD_SIZES[] = {{19, 19}, {19, 19}, {19, 19}, {13, 13}, {9, 9}, {9, 9}, {7, 11}, {19, 10}, {5, 5}, {4, 3}, {2, 2}, {13, 6}};
const float KOMIS[] = {7.5f, 6.5f, 7.0f, 6.0f, 0.5f, 0.0f, -0.5f, -7.5f, 5.5f, 3.0f, 150.0f, -150.0f, 0.75f, 400.0f, -17.25f};

struct Position {
  Board board;
  BoardHistory hist;
  Player pla;
  MiscNNInputParams params;
  int nnX, nnY;
};

MiscNNInputParams drawParams(Rand& rand) {
  MiscNNInputParams p;
  const double dews[] = {0.5, 0.5, 0.0, 1.0, 0.3};
  p.drawEquivalentWinsForWhite = dews[rand.nextUInt(5)];
  p.conservativePassAndIsRoot = rand.nextBool(0.35);
  p.enablePassingHacks = rand.nextBool(0.5);
  const double pdas[] = {0.0, 0.0, 1.5, -2.25, 0.01};
  p.playoutDoublingAdvantage = pdas[rand.nextUInt(5)];
  const int hists[] = {1000, 1000, 1000, 0, 1, 2, 3, 4, 5};
  p.maxHistory = hists[rand.nextUInt(9)];
  const int overrides[] = {-1, -1, 0, 1};
  p.passAliveSuicideRulesOverride = overrides[rand.nextUInt(4)];
  return p;
}

bool compareOne(const Position& q, long long index, bool verbose) {
  const int C = NNInputs::NUM_FEATURES_SPATIAL_V7, G = NNInputs::NUM_FEATURES_GLOBAL_V7;
  std::vector<float> refRow((size_t)C * q.nnX * q.nnY, -1.0f), ownRow(refRow.size(), -2.0f);
  float refGlobal[G], ownGlobal[G];
  std::vector<uint8_t> packed(KatamxFeatures::packedRowBytes(q.nnX, q.nnY), 0xAB);
  NNInputs::fillRowV7(q.board, q.hist, q.pla, q.params, q.nnX, q.nnY, true, refRow.data(), refGlobal);
  KatamxFeatures::fillPackedV7(q.board, q.hist, q.pla, q.params, q.nnX, q.nnY, packed.data(), ownGlobal);
  KatamxFeatures::unpackToNHWC(packed.data(), q.nnX, q.nnY, C, ownRow.data());
  bool same = memcmp(refRow.data(), ownRow.data(), refRow.size() * sizeof(float)) == 0 && memcmp(refGlobal, ownGlobal, sizeof(refGlobal)) == 0;
  // bits beyond the last cell of a plane must be zero (the device reads whole bytes)
  const int pb = KatamxFeatures::planeBytes(q.nnX, q.nnY), cells = q.nnX * q.nnY;
  for(int c = 0; c < C && (cells & 7); c++)
    if(packed[(size_t)c * pb + pb - 1] & (0xFFu >> (cells & 7)))
      same = false;
  if(!same && verbose) {
    printf("MISMATCH at position %lld: board %dx%d in %dx%d, pla %d, encore %d, moves %d, rules %s\n", index, q.board.x_size, q.board.y_size, q.nnX,
           q.nnY, (int)q.pla, q.hist.encorePhase, (int)q.hist.moveHistory.size(), q.hist.rules.toString().c_str());
    for(int p = 0; p < cells; p++)
      for(int c = 0; c < C; c++)
        if(refRow[(size_t)p * C + c] != ownRow[(size_t)p * C + c])
          printf("  plane %d cell (%d,%d): reference %g own %g\n", c, p % q.nnX, p / q.nnX, refRow[(size_t)p * C + c], ownRow[(size_t)p * C + c]);
    for(int g = 0; g < G; g++)
      if(memcmp(&refGlobal[g], &ownGlobal[g], 4) != 0)
        printf("  global %d: reference %.9g own %.9g\n", g, refGlobal[g], ownGlobal[g]);
  }
  return same;
}

}  // namespace

struct Tally {
  long long positions = 0, mismatches = 0, encore1 = 0, encore2 = 0, finished = 0, hidden = 0;
};

// One stream of random games (see the header of this file); kept: a sample of 19x19 positions for the timing run (may be NULL).
static Tally playAndCompare(int numGames, uint64_t seed, std::vector<Position>* keptOut) {
  Rand rand(seed);
  const bool timeIt = keptOut != NULL;
  std::vector<Position> keptLocal;
  std::vector<Position>& kept = keptOut != NULL ? *keptOut : keptLocal;
  long long positions = 0, mismatches = 0, encore1 = 0, encore2 = 0, finished = 0, hidden = 0;
  for(int game = 0; game < numGames; game++) {
    const Sizes bs = BOARD_SIZES[rand.nextUInt(sizeof(BOARD_SIZES) / sizeof(BOARD_SIZES[0]))];
    const bool exactLen = rand.nextBool(0.3);
    const int nnX = exactLen ? bs.x : NNPos::MAX_BOARD_LEN, nnY = exactLen ? bs.y : NNPos::MAX_BOARD_LEN;
    const int koRules[] = {Rules::KO_SIMPLE, Rules::KO_POSITIONAL, Rules::KO_SITUATIONAL, Rules::KO_SPIGHT};
    const int taxRules[] = {Rules::TAX_NONE, Rules::TAX_SEKI, Rules::TAX_ALL};
    const int bonusRules[] = {Rules::WHB_ZERO, Rules::WHB_N, Rules::WHB_N_MINUS_ONE};
    const int scoring = rand.nextBool(0.5) ? Rules::SCORING_AREA : Rules::SCORING_TERRITORY;
    // (the button exists under area scoring only, rules.cpp)
    const Rules rules(
      koRules[rand.nextUInt(4)], scoring, taxRules[rand.nextUInt(3)], rand.nextBool(0.4), scoring == Rules::SCORING_AREA && rand.nextBool(0.3),
      bonusRules[rand.nextUInt(3)], rand.nextBool(0.4), KOMIS[rand.nextUInt(sizeof(KOMIS) / sizeof(KOMIS[0]))]);
    Board board(bs.x, bs.y);
    const int area = bs.x * bs.y;
    Player pla = P_BLACK;
    if(area >= 25 && rand.nextBool(0.25)) {  // handicap stones
      const int n = 2 + (int)rand.nextUInt(4);
      for(int i = 0; i < n; i++) {
        const Loc loc = Location::getLoc((int)rand.nextUInt(bs.x), (int)rand.nextUInt(bs.y), bs.x);
        if(board.colors[loc] == C_EMPTY)
          board.setStone(loc, P_BLACK);
      }
      pla = P_WHITE;
    }
    BoardHistory hist(board, pla, rules, 0, rand.nextBool(0.3));
    hist.setAssumeMultipleStartingBlackMovesAreHandicap(rand.nextBool(0.5));
    const int maxMoves = area + area / 2 + 40;
    for(int move = 0; move <= maxMoves; move++) {
      pla = hist.presumedNextMovePla;
      // ---- compare at this position
      const int numParamSets = 2;
      for(int k = 0; k < numParamSets + 1; k++) {
        Position q{board, hist, k == numParamSets ? getOpp(pla) : pla, k == 0 ? MiscNNInputParams() : drawParams(rand), nnX, nnY};
        const bool same = compareOne(q, positions, mismatches < 5);
        positions++;
        mismatches += same ? 0 : 1;
        if(timeIt && kept.size() < 4000 && bs.x == 19 && bs.y == 19 && k == 0 && move % 7 == 3)
          kept.push_back(q);
      }
      encore1 += hist.encorePhase == 1;
      encore2 += hist.encorePhase == 2;
      hidden += hist.passWouldEndGame(board, pla);
      if(hist.isGameFinished) {
        finished++;
        break;
      }
      // ---- a random legal move; passes become likely late in the game
      std::vector<Loc> legal;
      for(int y = 0; y < bs.y; y++)
        for(int x = 0; x < bs.x; x++) {
          const Loc loc = Location::getLoc(x, y, bs.x);
          if(hist.isLegal(board, loc, pla))
            legal.push_back(loc);
        }
      const double passProb = move < area / 2 ? 0.02 : move < area ? 0.15 : 0.4;
      Loc loc = Board::PASS_LOC;
      if(!legal.empty() && !rand.nextBool(passProb))
        loc = legal[rand.nextUInt((uint32_t)legal.size())];
      if(!hist.isLegal(board, loc, pla))  // (a pass can be illegal in the encore: ko recapture blocked positions are handled by isLegal)
        break;
      hist.makeBoardMoveAssumeLegal(board, loc, pla, NULL);
    }
  }
  Tally t;
  t.positions = positions; t.mismatches = mismatches; t.encore1 = encore1; t.encore2 = encore2; t.finished = finished; t.hidden = hidden;
  return t;
}

// features_selftest <games> <seed> [time | threads N]: with `threads N`, N threads play <games> games each at the same time - the
// ladder memo is process-wide and shared by a search's threads, so a race in it would show as a wrong plane here.
int main(int argc, char** argv) {
  const int numGames = argc > 1 ? atoi(argv[1]) : 200;
  const uint64_t seed = argc > 2 ? strtoull(argv[2], NULL, 10) : 1;
  const bool timeIt = argc > 3 && strcmp(argv[3], "time") == 0;
  const int numThreads = argc > 4 && strcmp(argv[3], "threads") == 0 ? std::max(1, atoi(argv[4])) : 1;
  Board::initHash();
  ScoreValue::initTables();
  std::vector<Position> kept;
  std::vector<Tally> tallies(numThreads);
  if(numThreads == 1)
    tallies[0] = playAndCompare(numGames, seed, timeIt ? &kept : NULL);
  else {
    std::vector<std::thread> threads;
    for(int i = 0; i < numThreads; i++)
      threads.emplace_back([&tallies, i, numGames, seed] { tallies[i] = playAndCompare(numGames, seed + 1000003ULL * i, NULL); });
    for(std::thread& th : threads) th.join();
  }
  long long positions = 0, mismatches = 0, encore1 = 0, encore2 = 0, finished = 0, hidden = 0;
  for(const Tally& t : tallies) {
    positions += t.positions; mismatches += t.mismatches; encore1 += t.encore1; encore2 += t.encore2; finished += t.finished; hidden += t.hidden;
  }
  printf("features_selftest: %d games, %lld comparisons, %lld mismatches (positions in encore 1: %lld, encore 2: %lld, where a pass would end the game: %lld, finished games: %lld)\n",
         numGames * numThreads, positions, mismatches, encore1, encore2, hidden, finished);

  if(timeIt && !kept.empty()) {
    const int C = NNInputs::NUM_FEATURES_SPATIAL_V7;
    std::vector<float> row((size_t)C * 19 * 19);
    std::vector<uint8_t> packed(KatamxFeatures::packedRowBytes(19, 19));
    float gl[NNInputs::NUM_FEATURES_GLOBAL_V7];
    double best[2] = {1e30, 1e30};
    unsigned sink = 0;
    for(int rep = 0; rep < 5; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      for(const Position& q : kept) {
        NNInputs::fillRowV7(q.board, q.hist, q.pla, q.params, 19, 19, true, row.data(), gl);
        sink += (unsigned)row[5];
      }
      auto t1 = std::chrono::steady_clock::now();
      for(const Position& q : kept) {
        KatamxFeatures::fillPackedV7(q.board, q.hist, q.pla, q.params, 19, 19, packed.data(), gl);
        sink += packed[5];
      }
      auto t2 = std::chrono::steady_clock::now();
      best[0] = std::min(best[0], std::chrono::duration<double>(t1 - t0).count());
      best[1] = std::min(best[1], std::chrono::duration<double>(t2 - t1).count());
    }
    printf("timing over %zu 19x19 positions (best of 5): NNInputs::fillRowV7 %.2f us/row (fp32 row, before any packing), KatamxFeatures::fillPackedV7 %.2f us/row [%u]\n",
           kept.size(), best[0] / kept.size() * 1e6, best[1] / kept.size() * 1e6, sink);
  }
  return mismatches == 0 ? 0 : 1;
}
