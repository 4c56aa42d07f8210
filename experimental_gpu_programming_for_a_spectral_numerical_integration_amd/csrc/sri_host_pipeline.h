// sri_host_pipeline.h — how a HOST panel moves through the card (product code, pure C++: no HIP in this file).
//
// sri_apply_host / sri_integrate_host (include/sri_capi.h) are what the drop-in header calls when the reference's
// main.cpp hands over ordinary heap matrices (main.cpp:166-172: `Eigen::MatrixXd b_NN`, `r_stack`).  The library never
// gives such a pointer to the GPU runtime: no hipHostRegister / hipHostUnregister on caller memory, no asynchronous copy
// from or to it.  The caller's memory is only ever touched by the CPU (memcpy, by the threads below); what the copy
// engines see are pinned staging slots that the handle owns from its first host call to its destruction.  (Round 3
// page-locked the caller's buffers for the duration of each call instead, and two runs of the GPU suite ended in a GPU
// memory fault on a host address — DESIGN.md §4 "Host panels" has the record; this form has nothing left to unlock.)
//
// The panel is cut into blocks of whole column tiles; block b goes through
//   IN(b)   caller X (any leading dimension) -> pinned slot, compacted to ld = rows     CPU, CopyCrew
//   UP(b)   pinned slot -> device slot                                                   copy engine, stream "in"
//   RUN(b)  the apply on the device slot                                                 stream "run"
//   DOWN(b) device slot -> pinned slot                                                   copy engine, stream "out"
//   OUT(b)  pinned slot -> caller Y (any leading dimension)                              CPU, CopyCrew
// with IN(b+1), UP/RUN/DOWN of the blocks in between and OUT(b-2) all running at the same time.  The driver below is a
// template over the device side, so that tests/cpp/host_pipeline_check.cpp can run it against a mock "card" made of
// threads (under ThreadSanitizer) and check every ordering the real one relies on, without a GPU.
#ifndef SRI_HOST_PIPELINE_H
#define SRI_HOST_PIPELINE_H

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace sri {

// A group of copy tasks the submitter can wait for.
struct CopyJob {
  std::atomic<size_t> remaining{0};
  bool idle() const { return remaining.load(std::memory_order_acquire) == 0; }
};

// Persistent worker threads that do nothing but memcpy: `count` runs of `width` bytes, `dst_stride` / `src_stride` apart.
// The submitting thread helps while it waits, so a crew of zero threads is simply a synchronous copy.
class CopyCrew {
 public:
  explicit CopyCrew(unsigned threads) {
    try {
      for (unsigned i = 0; i < threads; ++i) workers_.emplace_back([this] { work(); });
    } catch (...) {                       // fewer workers than asked for (or none): still correct, the submitter helps
    }
  }
  ~CopyCrew() {
    {
      std::lock_guard<std::mutex> lock(mutex_);
      closing_ = true;
    }
    work_cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  CopyCrew(const CopyCrew&) = delete;
  CopyCrew& operator=(const CopyCrew&) = delete;
  unsigned threads() const { return static_cast<unsigned>(workers_.size()); }

  // Queue the copy as tasks of about `grain` bytes.  May throw std::bad_alloc (nothing is queued then).
  void submit(CopyJob& job, char* dst, size_t dst_stride, const char* src, size_t src_stride, size_t width, size_t count,
              size_t grain = size_t(1) << 20) {
    if (width == 0 || count == 0) return;
    std::vector<Task> tasks;
    if (count == 1 || (dst_stride == width && src_stride == width)) {        // one linear run: cut it into pieces
      const size_t total = width * count;
      for (size_t o = 0; o < total; o += grain) tasks.push_back(Task{dst + o, 0, src + o, 0, std::min(grain, total - o), 1, &job});
    } else {                                                                  // whole runs per task
      const size_t per = std::max<size_t>(grain / width, 1);
      for (size_t c = 0; c < count; c += per)
        tasks.push_back(Task{dst + c * dst_stride, dst_stride, src + c * src_stride, src_stride, width, std::min(per, count - c), &job});
    }
    {
      std::lock_guard<std::mutex> lock(mutex_);
      if (head_ == queue_.size()) { queue_.clear(); head_ = 0; }
      queue_.reserve(queue_.size() + tasks.size());                          // the only thing that can throw: nothing queued yet
      job.remaining.fetch_add(tasks.size(), std::memory_order_relaxed);
      queue_.insert(queue_.end(), tasks.begin(), tasks.end());
    }
    work_cv_.notify_all();
  }

  // Returns when every task submitted under `job` has run; runs queued tasks (of any job) itself meanwhile.
  void wait(CopyJob& job) {
    std::unique_lock<std::mutex> lock(mutex_);
    while (job.remaining.load(std::memory_order_acquire) != 0) {
      if (head_ < queue_.size()) {
        const Task t = queue_[head_++];
        lock.unlock();
        run(t);
        lock.lock();
      } else {
        done_cv_.wait(lock);
      }
    }
  }

 private:
  struct Task {
    char* dst;
    size_t dst_stride;
    const char* src;
    size_t src_stride;
    size_t width, count;
    CopyJob* job;
  };
  void run(const Task& t) {
#ifdef SRI_COPY_CREW_TEST_HOOK
    SRI_COPY_CREW_TEST_HOOK();          // tests/cpp/host_pipeline_check.cpp stalls random tasks here; never defined in the product build
#endif
    for (size_t i = 0; i < t.count; ++i) std::memcpy(t.dst + i * t.dst_stride, t.src + i * t.src_stride, t.width);
    if (t.job->remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) {
      std::lock_guard<std::mutex> lock(mutex_);                              // pairs with the predicate check in wait()
      done_cv_.notify_all();
    }
  }
  void work() {
    std::unique_lock<std::mutex> lock(mutex_);
    for (;;) {
      work_cv_.wait(lock, [this] { return closing_ || head_ < queue_.size(); });
      if (head_ == queue_.size()) return;                                    // closing, nothing left
      const Task t = queue_[head_++];
      lock.unlock();
      run(t);
      lock.lock();
    }
  }
  std::mutex mutex_;
  std::condition_variable work_cv_, done_cv_;
  std::vector<Task> queue_;                                                 // FIFO: [head_, size())
  size_t head_ = 0;
  std::vector<std::thread> workers_;
  bool closing_ = false;
};

// Columns per block: 4..32 MiB (the link is at 90 % of its rate from 4 MiB on, 98 % at 32 MiB), at least ~8 blocks per
// call so that the stages overlap, whole 128-column tiles.
inline size_t host_block_columns(size_t rows, size_t M) {
  const size_t col_bytes = rows * sizeof(double);
  const size_t total = col_bytes * M;
  size_t target = total / 8;
  target = std::min<size_t>(std::max<size_t>(target, size_t(4) << 20), size_t(32) << 20);
  size_t cols = std::max<size_t>(target / col_bytes, 1);
  cols = (cols + 127) & ~size_t(127);
  return std::min(cols, (M + 127) & ~size_t(127));
}

// Byte layout of a staging / device slot holding `cols` columns: the compacted panel block (ld = rows), then — for
// sri_integrate_host — the block's slice of x_base, 16-byte aligned.
inline size_t slot_base_offset(size_t rows, size_t cols) { return (rows * cols * sizeof(double) + 15) & ~size_t(15); }
inline size_t slot_bytes(size_t rows, size_t cols, bool with_base) {
  return slot_base_offset(rows, cols) + (with_base ? cols * sizeof(double) : 0);
}

// The driver.  `Card` provides (int results are sri_status codes, 0 = ok):
//   static constexpr int SLOTS                       ring depth (>= 3)
//   char* stage_in(int slot), char* stage_out(int slot)      pinned staging, >= slot_bytes(rows, block columns, with_base)
//   int enqueue(int slot, size_t c0, size_t cols, size_t in_bytes, size_t out_bytes, size_t base_offset_or_0)
//                                                    UP + RUN + DOWN of one block, asynchronous; the card orders its own
//                                                    slot reuse (device-side), the driver orders the staging slots
//   int wait_uploaded(int slot)                      UP of the block last enqueued on `slot` has completed
//   int wait_downloaded(int slot)                    DOWN ... has completed
//   void drain()                                     after an error: nothing of this call may still be in flight on return
// X / Y: column-major rows x M with leading dimensions ldx / ldy (doubles); Y may be X (in place: block b is read by IN(b)
// before OUT(b) writes it, and blocks are disjoint).  x_base: M doubles or nullptr.
template <class Card>
int run_host_pipeline(Card& card, CopyCrew& crew, const double* X, size_t ldx, const double* x_base, double* Y, size_t ldy,
                      size_t rows, size_t M, size_t block_cols) {
  constexpr int S = Card::SLOTS;
  static_assert(S >= 3, "IN(b), the blocks on the card and OUT(b-2) need three staging slots");
  constexpr size_t DEPTH = 2;                               // OUT(b - DEPTH) is retired while IN(b) runs
  const size_t col_bytes = rows * sizeof(double);
  const size_t blocks = (M + block_cols - 1) / block_cols;
  CopyJob in_job, out_job[S];
  int rc = 0;
  auto cols_of = [&](size_t b) { return std::min(block_cols, M - b * block_cols); };
  // hand block b's result back to the caller (asynchronously: the crew copies while the driver goes on)
  auto retire = [&](size_t b) -> int {
    const int slot = static_cast<int>(b % S);
    const int e = card.wait_downloaded(slot);
    if (e != 0) return e;
    crew.submit(out_job[slot], reinterpret_cast<char*>(Y + b * block_cols * ldy), ldy * sizeof(double), card.stage_out(slot), col_bytes,
                col_bytes, cols_of(b));
    return 0;
  };
  auto abandon = [&](int code) {
    crew.wait(in_job);
    for (int s = 0; s < S; ++s) crew.wait(out_job[s]);
    card.drain();
    return code;
  };
  try {
    for (size_t b = 0; b < blocks; ++b) {
      const int slot = static_cast<int>(b % S);
      const size_t c0 = b * block_cols, cols = cols_of(b);
      // stage_in(slot) is free once the upload of block b - S has left it (implied by retire(b - 1 - DEPTH) of the previous
      // iteration as long as DEPTH + 1 <= S; kept explicit so that the ring depth and DEPTH can change independently)
      if (b >= static_cast<size_t>(S) && (rc = card.wait_uploaded(slot)) != 0) return abandon(rc);
      crew.submit(in_job, card.stage_in(slot), col_bytes, reinterpret_cast<const char*>(X + c0 * ldx), ldx * sizeof(double), col_bytes, cols);
      const size_t base_off = x_base != nullptr ? slot_base_offset(rows, cols) : 0;
      if (x_base != nullptr)
        crew.submit(in_job, card.stage_in(slot) + base_off, cols * sizeof(double), reinterpret_cast<const char*>(x_base + c0), cols * sizeof(double),
                    cols * sizeof(double), 1);
      crew.wait(in_job);                                      // (the calling thread copies too)
      // stage_out(slot) is free once OUT(b - S) has copied it to the caller
      crew.wait(out_job[slot]);
      rc = card.enqueue(slot, c0, cols, slot_bytes(rows, cols, x_base != nullptr), cols * col_bytes, base_off);
      if (rc != 0) return abandon(rc);
      // Only now wait for the card: block b is on its way before the driver blocks on DOWN(b - DEPTH), which ends about when
      // UP(b - 1) does — so UP(b) follows UP(b - 1) without a gap, and IN(b + 1) has a whole block time to finish under it.
      // (Retiring before the enqueue, as the first version did, put this wait and a launch latency between any two uploads:
      // 13.8 ms instead of 12.8 ms per 0.54 GB panel at N = 256, profiles/r04_host_panel_bench.jsonl.)
      if (b >= DEPTH && (rc = retire(b - DEPTH)) != 0) return abandon(rc);
    }
    for (size_t b = blocks > DEPTH ? blocks - DEPTH : 0; b < blocks; ++b)
      if ((rc = retire(b)) != 0) return abandon(rc);
  } catch (const std::bad_alloc&) {
    return abandon(-2 /* SRI_ENOMEM */);
  }
  for (int s = 0; s < S; ++s) crew.wait(out_job[s]);
  return 0;
}

}  // namespace sri

#endif  // SRI_HOST_PIPELINE_H
