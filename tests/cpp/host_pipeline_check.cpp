// host_pipeline_check.cpp — the host-panel pipeline (csrc/sri_host_pipeline.h) against a MOCK card made of threads.
//
// Test infrastructure, CPU only (built by tests/test_host_pipeline.py, plain and under ThreadSanitizer).  The mock keeps
// the real card's structure — three in-order "streams" (up / run / down) whose slot reuse is ordered by per-slot event
// counters, random delays everywhere — and computes a result that depends on the GLOBAL column index, so a block that
// lands in the wrong place, a staging slot overwritten before its upload left it, or a result copied out before its
// download arrived shows up as a wrong value (and, under TSan, as a data race).
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <limits>
#include <random>

// random stalls inside the copy threads' tasks: without them a copy task practically never outlives the driver's next steps
// and an ordering the driver forgot (say, reusing a staging slot that is still being copied out) would go unnoticed
static void crew_task_stall();
#define SRI_COPY_CREW_TEST_HOOK crew_task_stall
#include "sri_host_pipeline.h"

static std::atomic<unsigned> g_stall_state{99991};
static std::atomic<unsigned> g_stall_us{0};
static void crew_task_stall() {
  const unsigned limit = g_stall_us.load(std::memory_order_relaxed);
  if (limit == 0) return;
  const unsigned r = g_stall_state.fetch_add(2246822519u) >> 7;
  if (r % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(r % limit));
}

namespace {

struct Stream {                                    // an in-order queue of closures on its own thread
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool closing = false;
  std::thread t;
  Stream() : t([this] { loop(); }) {}
  ~Stream() {
    {
      std::lock_guard<std::mutex> l(m);
      closing = true;
    }
    cv.notify_all();
    t.join();
  }
  void push(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> l(m);
      q.push_back(std::move(f));
    }
    cv.notify_all();
  }
  void loop() {
    std::unique_lock<std::mutex> l(m);
    for (;;) {
      cv.wait(l, [this] { return closing || !q.empty(); });
      if (q.empty()) return;
      auto f = std::move(q.front());
      q.pop_front();
      l.unlock();
      f();
      l.lock();
    }
  }
};

struct Event {                                     // "recorded up to sequence number n"
  std::mutex m;
  std::condition_variable cv;
  size_t done = 0;
  void signal(size_t n) {
    {
      std::lock_guard<std::mutex> l(m);
      done = n;
    }
    cv.notify_all();
  }
  void wait(size_t n) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return done >= n; });
  }
};

struct MockCard {
  static constexpr int SLOTS = 3;
  size_t rows;
  bool with_base;
  std::vector<char> in[SLOTS], out[SLOTS], dX[SLOTS], dY[SLOTS];
  Event landed[SLOTS], applied[SLOTS], returned[SLOTS];
  size_t last_seq[SLOTS] = {0, 0, 0};              // sequence number (block index + 1) last enqueued on the slot
  size_t prev_seq[SLOTS] = {0, 0, 0};
  size_t next_seq = 1;
  int fail_at = -1;                                // enqueue number that reports an error (negative: never)
  int enqueued = 0;
  unsigned jitter_us;
  std::atomic<unsigned> rng_state{12345};
  Stream up, run, down;                            // declared last: joined first on destruction

  MockCard(size_t rows_, size_t block_cols, bool base, unsigned jitter) : rows(rows_), with_base(base), jitter_us(jitter) {
    for (int s = 0; s < SLOTS; ++s) {
      in[s].assign(sri::slot_bytes(rows, block_cols, base), char(0x7f));
      out[s].assign(rows * block_cols * 8, char(0x7f));
      dX[s].assign(in[s].size(), 0);
      dY[s].assign(out[s].size(), 0);
    }
  }
  void nap() {
    if (jitter_us == 0) return;
    const unsigned r = rng_state.fetch_add(2654435761u) >> 8;
    std::this_thread::sleep_for(std::chrono::microseconds(r % jitter_us));
  }
  char* stage_in(int slot) { return in[slot].data(); }
  char* stage_out(int slot) { return out[slot].data(); }
  int enqueue(int slot, size_t c0, size_t cols, size_t in_bytes, size_t out_bytes, size_t base_off) {
    if (enqueued++ == fail_at) return -7;
    if (in_bytes > in[slot].size() || out_bytes > out[slot].size() || out_bytes != cols * rows * 8) std::abort();
    if (with_base != (base_off != 0)) std::abort();
    // from here until `returned` the out slot belongs to the card, which may write it at any time: do so at once, so that a
    // driver that has not finished copying the slot's PREVIOUS block out hands garbage to the caller
    std::memset(out[slot].data(), 0x55, out[slot].size());
    const size_t seq = next_seq++, before = last_seq[slot];
    prev_seq[slot] = before;
    last_seq[slot] = seq;
    up.push([=] {
      applied[slot].wait(before);                  // the apply that read dX[slot] is done
      nap();
      std::memcpy(dX[slot].data(), in[slot].data(), in_bytes);
      landed[slot].signal(seq);
    });
    run.push([=] {
      landed[slot].wait(seq);
      returned[slot].wait(before);                 // the download that read dY[slot] is done
      nap();
      const double* x = reinterpret_cast<const double*>(dX[slot].data());
      const double* xb = base_off ? reinterpret_cast<const double*>(dX[slot].data() + base_off) : nullptr;
      double* y = reinterpret_cast<double*>(dY[slot].data());
      for (size_t j = 0; j < cols; ++j)
        for (size_t i = 0; i < rows; ++i)
          y[j * rows + i] = 2.0 * x[j * rows + i] + static_cast<double>(c0 + j) + 0.5 * static_cast<double>(i) - (xb ? xb[j] : 0.0);
      applied[slot].signal(seq);
    });
    down.push([=] {
      applied[slot].wait(seq);
      nap();
      std::memcpy(out[slot].data(), dY[slot].data(), out_bytes);
      returned[slot].signal(seq);
    });
    return 0;
  }
  int wait_uploaded(int slot) {
    landed[slot].wait(last_seq[slot]);
    return 0;
  }
  int wait_downloaded(int slot) {
    returned[slot].wait(last_seq[slot]);
    return 0;
  }
  bool drained = false;
  void drain() {
    for (int s = 0; s < SLOTS; ++s) returned[s].wait(last_seq[s]);
    drained = true;
  }
};

int failures = 0;
#define CHECK(cond, ...)                         \
  do {                                           \
    if (!(cond)) {                               \
      ++failures;                                \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);                  \
      std::printf("\n");                         \
    }                                            \
  } while (0)

void one_case(size_t rows, size_t M, size_t ldx, size_t ldy, bool base, bool in_place, unsigned threads, size_t block_cols, unsigned jitter,
              unsigned seed) {
  const double nan = std::numeric_limits<double>::quiet_NaN();
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  if (in_place) ldy = ldx;
  std::vector<double> X(ldx * M + 3, nan), Y(ldy * M + 3, -7.25), xb(M);
  for (size_t j = 0; j < M; ++j) {
    xb[j] = u(rng);
    for (size_t i = 0; i < rows; ++i) X[j * ldx + i] = u(rng);
  }
  const std::vector<double> X0 = X;
  double* y = in_place ? X.data() : Y.data();
  sri::CopyCrew crew(threads);
  MockCard card(rows, block_cols, base, jitter);
  g_stall_us.store(threads != 0 ? 4 * jitter : 0);
  const int rc = sri::run_host_pipeline(card, crew, X.data(), ldx, base ? xb.data() : nullptr, y, ldy, rows, M, block_cols);
  g_stall_us.store(0);
  CHECK(rc == 0, "rc %d", rc);
  size_t bad = 0;
  for (size_t j = 0; j < M; ++j) {
    for (size_t i = 0; i < rows; ++i) {
      const double want = 2.0 * X0[j * ldx + i] + static_cast<double>(j) + 0.5 * static_cast<double>(i) - (base ? xb[j] : 0.0);
      if (y[j * ldy + i] != want) ++bad;
    }
    for (size_t i = rows; i < ldy && j + 1 < M; ++i) {            // padding rows between columns: untouched
      const double pad = y[j * ldy + i];
      if (in_place ? !std::isnan(pad) : pad != -7.25) ++bad;
    }
  }
  if (!in_place)
    for (size_t k = 0; k < X.size(); ++k)
      if (!(X[k] == X0[k] || (std::isnan(X[k]) && std::isnan(X0[k])))) ++bad;   // the input is read-only
  CHECK(bad == 0, "rows %zu M %zu ldx %zu ldy %zu base %d inplace %d threads %u block %zu: %zu wrong values", rows, M, ldx, ldy, int(base),
        int(in_place), threads, block_cols, bad);
}

void failing_card() {
  // an error from the card in the middle of the panel: the driver reports it and leaves nothing in flight
  const size_t rows = 16, M = 2000, cb = 128;
  std::vector<double> X(rows * M, 1.0), Y(rows * M, 0.0);
  sri::CopyCrew crew(2);
  MockCard card(rows, cb, false, 50);
  card.fail_at = 5;
  const int rc = sri::run_host_pipeline(card, crew, X.data(), rows, nullptr, Y.data(), rows, rows, M, cb);
  CHECK(rc == -7, "rc %d", rc);
  CHECK(card.drained, "the card was not drained after an error");
}

void crew_alone() {
  // the crew by itself: strided and linear copies, several jobs in flight, zero workers
  for (unsigned threads : {0u, 1u, 5u}) {
    sri::CopyCrew crew(threads);
    const size_t w = 1000, n = 777, ss = 1024, ds = 1500;
    std::vector<char> src(ss * n), dst(ds * n, 0), lin(3 << 20), lin2(3 << 20, 0);
    for (size_t i = 0; i < src.size(); ++i) src[i] = static_cast<char>(i * 31 + 7);
    for (size_t i = 0; i < lin.size(); ++i) lin[i] = static_cast<char>(i * 17 + 3);
    sri::CopyJob a, b;
    crew.submit(a, dst.data(), ds, src.data(), ss, w, n, 64 << 10);
    crew.submit(b, lin2.data(), lin.size(), lin.data(), lin.size(), lin.size(), 1, 256 << 10);
    crew.wait(b);
    crew.wait(a);
    CHECK(a.idle() && b.idle(), "jobs not idle");
    size_t bad = 0;
    for (size_t c = 0; c < n; ++c)
      for (size_t i = 0; i < ds; ++i) bad += dst[c * ds + i] != (i < w ? src[c * ss + i] : 0);
    CHECK(bad == 0 && lin == lin2, "crew(%u): %zu wrong bytes", threads, bad);
  }
}

}  // namespace

int main(int argc, char** argv) {
  const bool quick = argc > 1 && std::string(argv[1]) == "quick";        // the TSan run: fewer, smaller cases
  crew_alone();
  failing_card();
  unsigned seed = 1;
  const size_t shapes[][2] = {{1, 1}, {1, 300}, {16, 127}, {63, 128}, {255, 129}, {256, 1000}, {64, 5000}, {255, 2600}, {300, 385}};
  for (const auto& sh : shapes) {
    const size_t rows = sh[0], M = sh[1];
    if (quick && rows * M > 300000) continue;
    for (unsigned threads : {0u, 1u, 4u})
      for (int variant = 0; variant < 6; ++variant) {
        const bool base = variant & 1, in_place = variant == 4, strided = variant >= 2;
        const size_t ldx = rows + (strided ? 3 : 0), ldy = rows + (strided && variant != 3 ? 5 : 0);
        // block sizes: the product's own choice, and a small one that makes many blocks (>= 3 rounds of the slot ring)
        for (size_t cb : {sri::host_block_columns(rows, M), size_t(128)}) {
          if (quick && threads == 1) continue;
          one_case(rows, M, ldx, ldy, base, in_place, threads, cb, quick ? 30 : 200, seed++);
        }
      }
  }
  // block-size rule (what the product uses): whole 128-column tiles, 4..32 MiB, never more than the padded panel
  for (size_t rows : {1, 15, 64, 255, 256, 1024})
    for (size_t M : {1, 128, 129, 4096, 262144, 1048576}) {
      const size_t cb = sri::host_block_columns(rows, M);
      CHECK(cb % 128 == 0 && cb >= 128 && cb <= ((M + 127) & ~size_t(127)), "block columns %zu for rows %zu M %zu", cb, rows, M);
      CHECK(cb * rows * 8 <= (size_t(32) << 20) + 128 * rows * 8, "block of %zu bytes", cb * rows * 8);
    }
  std::printf("host_pipeline_check: %d failures\n", failures);
  return failures == 0 ? 0 : 1;
}
