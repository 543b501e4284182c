// leaf_pump.cpp — a caller of the persistent leaf batcher as a search would be (measured from the
// boundary): T threads, each keeping K leaves in flight through kmx_batcher_submit / kmx_batcher_wait — tickets instead of
// one blocked OS thread per leaf, which is what bounds the reference's own search (cpp/search/search.cpp:1189-1463 evaluates a
// leaf with a blocking NNEvaluator::evaluate, nneval.cpp:861-936). Uses include/katamx.h only; rows are synthetic 0/1 feature
// planes. Prints one JSON line: rows per second through the C ABI from HOST rows (packing, H2D, pass, D2H, delivery included).
//   leaf_pump <model.bin[.gz]> <nn_len> <max_batch> <max_in_flight> <threads> <tickets_per_thread> <seconds> [precision: auto|bf16|fp16]
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "katamx.h"

static void die(const char* what) {
  fprintf(stderr, "leaf_pump: %s: %s\n", what, kmx_last_error());
  exit(1);
}

int main(int argc, char** argv) {
  if(argc < 8) {
    fprintf(stderr, "usage: leaf_pump model nn_len max_batch max_in_flight threads tickets_per_thread seconds [precision]\n");
    return 2;
  }
  const char* modelPath = argv[1];
  const int L = atoi(argv[2]), maxBatch = atoi(argv[3]), inFlight = atoi(argv[4]), T = atoi(argv[5]), K = atoi(argv[6]);
  const double seconds = atof(argv[7]);
  const std::string prec = argc > 8 ? argv[8] : "auto";
  const int precision = prec == "fp16" ? KMX_PREC_FP16 : prec == "bf16" ? KMX_PREC_BF16 : KMX_PREC_AUTO;
  if(kmx_global_init() != KMX_OK) die("kmx_global_init");
  kmx_model* model = nullptr;
  if(kmx_model_load(modelPath, "", &model) != KMX_OK) die("kmx_model_load");
  kmx_model_info info;
  if(kmx_model_info_get(model, &info) != KMX_OK) die("kmx_model_info_get");
  const int gpus[1] = {0};
  kmx_context* ctx = nullptr;
  if(kmx_context_create(gpus, 1, L, L, precision, &ctx) != KMX_OK) die("kmx_context_create");
  kmx_batcher* b = nullptr;
  if(kmx_batcher_create(ctx, model, maxBatch, inFlight, 0, &b) != KMX_OK) die("kmx_batcher_create");

  const int S = L * L, C = info.num_input_channels, G = info.num_input_global_channels;
  const int POOL = 64;  // distinct synthetic positions
  std::vector<float> spatial((size_t)POOL * S * C, 0.0f), global((size_t)POOL * G, 0.0f);
  uint32_t rng = 20260921u;
  auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  for(int p = 0; p < POOL; p++)
    for(int i = 0; i < S; i++) {
      float* cell = &spatial[((size_t)p * S + i) * C];
      cell[0] = 1.0f;  // on board
      const unsigned r = next() % 10;
      if(r < 2) cell[1] = 1.0f;
      else if(r < 4) cell[2] = 1.0f;
      for(int c = 3; c < C; c++) cell[c] = (next() % 20 == 0) ? 1.0f : 0.0f;
    }
  for(float& g : global) g = (float)(next() % 1000) / 1000.0f - 0.5f;

  std::atomic<bool> stop(false);
  std::atomic<uint64_t> done(0);
  std::atomic<int> failed(0);
  auto worker = [&](int id) {
    struct Leaf {
      uint64_t ticket;
      std::vector<float> policy, own;
      float value[3], score[6];
    };
    std::vector<Leaf> leaves(K);
    for(Leaf& l : leaves) {
      l.policy.resize(S + 1);
      l.own.resize(S);
    }
    std::deque<int> pending;
    std::vector<int> freeSlots;
    for(int i = 0; i < K; i++) freeSlots.push_back(i);
    uint32_t r = 777u + (uint32_t)id * 7919u;
    while(true) {
      const bool stopping = stop.load(std::memory_order_relaxed);
      while(!stopping && !freeSlots.empty()) {
        const int s = freeSlots.back();
        freeSlots.pop_back();
        r = r * 1664525u + 1013904223u;
        const int p = (int)((r >> 8) % POOL);
        if(kmx_batcher_submit(b, &spatial[(size_t)p * S * C], &global[(size_t)p * G], nullptr, (int)((r >> 4) & 7), 0.0f, leaves[s].policy.data(),
                              leaves[s].value, leaves[s].score, (r & 1) ? leaves[s].own.data() : nullptr, &leaves[s].ticket) != KMX_OK) {
          failed++;
          return;
        }
        pending.push_back(s);
      }
      if(pending.empty()) break;
      const int s = pending.front();
      pending.pop_front();
      if(kmx_batcher_wait(b, leaves[s].ticket) != KMX_OK) {
        failed++;
        return;
      }
      done.fetch_add(1, std::memory_order_relaxed);
      freeSlots.push_back(s);
    }
  };
  std::vector<std::thread> threads;
  for(int i = 0; i < T; i++) threads.emplace_back(worker, i);
  std::this_thread::sleep_for(std::chrono::milliseconds(500));  // warm-up
  uint64_t r0 = 0, b0 = 0, r1 = 0, b1 = 0;
  kmx_batcher_stats(b, &r0, &b0);
  const auto t0 = std::chrono::steady_clock::now();
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  kmx_batcher_stats(b, &r1, &b1);
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  stop = true;
  for(std::thread& t : threads) t.join();
  printf("{\"rows_per_s\": %.1f, \"batches_per_s\": %.1f, \"avg_batch\": %.1f, \"threads\": %d, \"tickets_per_thread\": %d, \"max_batch\": %d, "
         "\"max_in_flight\": %d, \"seconds\": %.2f, \"failed\": %d}\n",
         (double)(r1 - r0) / el, (double)(b1 - b0) / el, b1 > b0 ? (double)(r1 - r0) / (double)(b1 - b0) : 0.0, T, K, maxBatch, inFlight, el, failed.load());
  kmx_batcher_free(b);
  kmx_context_free(ctx);
  kmx_model_free(model);
  return failed.load() ? 1 : 0;
}
