// sri_host_panels.hip — sri_apply_host / sri_integrate_host: the apply for callers that hold HOST panels.
//
// This is what include/chebyshev_differentiation.h calls for the reference's main.cpp, whose matrices are ordinary heap
// memory (main.cpp:166-172: `Eigen::MatrixXd b_NN`, `r_stack = Dn_NN_inv*(b_NN - ivp)`).  The panel crosses PCIe, and the
// link is the whole cost (0.54 GB each way at N = 256, 65 536 rods against 0.5 ms of compute), so the call is a pipeline of
// column blocks: CPU threads copy block b+1 into a pinned staging slot while the copy engines carry block b to the card,
// the kernel runs block b-1, and blocks b-2, b-3 travel back and are copied out (sri_host_pipeline.h has the driver; the
// stage order is proved there against a mock card under ThreadSanitizer, tests/cpp/host_pipeline_check.cpp).
//
// What this file guarantees by construction (DESIGN.md §4 "Host panels" has the history: round 3 page-locked the caller's
// buffers per call and saw two GPU memory faults on host addresses):
//   * no hipHostRegister / hipHostUnregister, ever;
//   * no HIP call is handed a pointer into the caller's X, Y or x_base — unless the caller declared them page-locked
//     itself (sri_apply_set_host_memory_pinned), in which case their lifetime and locking are the caller's and the
//     library only checks the claim (hipPointerGetAttributes) and copies straight from / to them;
//   * everything the copy engines touch on the host side is allocated with hipHostMalloc at the handle's first host call
//     (or when a larger panel arrives) and freed with the handle, after the three streams have drained.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <new>

#define SRI_DEVICE_HELPERS 1
#include "sri_host_pipeline.h"
#include "sri_internal.h"

namespace sri {

namespace {

struct PanelDeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t enter(int dev) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev != dev) {
      e = hipSetDevice(dev);
      if (e != hipSuccess) return e;
      switched = true;
    }
    return hipSuccess;
  }
  ~PanelDeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

unsigned default_copy_threads() {
  // Four threads plus the calling one keep both directions of the link fed on the GPU box's host (12.8 ms per 0.54 GB panel
  // against 12.0 ms straight from page-locked memory; 2 threads: 17.2, none: 40.9; more than 4 buy nothing —
  // profiles/r04_host_panel_bench.jsonl), and a node runs one handle per GPU.
  const unsigned hw = std::thread::hardware_concurrency();
  return hw == 0 ? 4u : std::min(4u, hw);
}

}  // namespace

struct HostPipe {
  static constexpr int SLOTS = 3;
  hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
  hipEvent_t landed[SLOTS] = {}, applied[SLOTS] = {}, returned[SLOTS] = {};
  bool used[SLOTS] = {};                                        // the slot's events have been recorded at least once
  char* dX[SLOTS] = {};                                         // device: block of X (+ its slice of x_base)
  char* dY[SLOTS] = {};
  char* hIn[SLOTS] = {};                                        // pinned host staging, library-owned
  char* hOut[SLOTS] = {};
  size_t dev_in_bytes = 0, dev_out_bytes = 0, host_in_bytes = 0, host_out_bytes = 0;
  CopyCrew crew;

  explicit HostPipe(unsigned threads) : crew(threads) {}

  int hip_fail(hipError_t e) {
    record_hip_error(static_cast<int>(e));
    (void)hipGetLastError();
    return SRI_EHIP;
  }
  int init() {
    if (s_in != nullptr) return SRI_OK;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking)) != hipSuccess) return hip_fail(e);
    if ((e = hipStreamCreateWithFlags(&s_run, hipStreamNonBlocking)) != hipSuccess) return hip_fail(e);
    if ((e = hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking)) != hipSuccess) return hip_fail(e);
    for (int i = 0; i < SLOTS; ++i) {
      if ((e = hipEventCreateWithFlags(&landed[i], hipEventDisableTiming)) != hipSuccess) return hip_fail(e);
      if ((e = hipEventCreateWithFlags(&applied[i], hipEventDisableTiming)) != hipSuccess) return hip_fail(e);
      if ((e = hipEventCreateWithFlags(&returned[i], hipEventDisableTiming)) != hipSuccess) return hip_fail(e);
    }
    return SRI_OK;
  }
  // (re)size one family of SLOTS buffers; nothing is in flight between calls (every call drains before it returns).  The larger
  // buffers are allocated FIRST and swapped in only when all of them exist: a failed grow leaves the handle with the (smaller)
  // buffers it had, so that a later call that fits them still works.  The pipe never shrinks by itself (sri_apply_host_trim).
  template <class Alloc, class Free>
  int grow(char* (&bufs)[SLOTS], size_t& have, size_t want, Alloc alloc, Free release) {
    if (want <= have) return SRI_OK;
    char* fresh[SLOTS] = {};
    for (int i = 0; i < SLOTS; ++i) {
      if (alloc(&fresh[i], want) != hipSuccess) {
        (void)hipGetLastError();
        for (int j = 0; j < i; ++j) (void)release(fresh[j]);
        return SRI_ENOMEM;
      }
    }
    for (int i = 0; i < SLOTS; ++i) {
      if (bufs[i]) (void)release(bufs[i]);
      bufs[i] = fresh[i];
    }
    have = want;
    return SRI_OK;
  }
  // give back the staging ring and the device slots (streams and events stay); the next host call allocates what it needs
  void trim() {
    drain();
    for (int i = 0; i < SLOTS; ++i) {
      if (dX[i]) (void)hipFree(dX[i]);
      if (dY[i]) (void)hipFree(dY[i]);
      if (hIn[i]) (void)hipHostFree(hIn[i]);
      if (hOut[i]) (void)hipHostFree(hOut[i]);
      dX[i] = dY[i] = hIn[i] = hOut[i] = nullptr;
    }
    dev_in_bytes = dev_out_bytes = host_in_bytes = host_out_bytes = 0;
  }
  int reserve(size_t in_bytes, size_t out_bytes, bool staging) {
    int rc = init();
    auto dev_alloc = [](char** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); };
    auto dev_free = [](char* p) { return hipFree(p); };
    auto pin_alloc = [](char** p, size_t n) { return hipHostMalloc(reinterpret_cast<void**>(p), n, hipHostMallocDefault); };
    auto pin_free = [](char* p) { return hipHostFree(p); };
    if (rc == SRI_OK) rc = grow(dX, dev_in_bytes, in_bytes, dev_alloc, dev_free);
    if (rc == SRI_OK) rc = grow(dY, dev_out_bytes, out_bytes, dev_alloc, dev_free);
    if (rc == SRI_OK && staging) rc = grow(hIn, host_in_bytes, in_bytes, pin_alloc, pin_free);
    if (rc == SRI_OK && staging) rc = grow(hOut, host_out_bytes, out_bytes, pin_alloc, pin_free);
    return rc;
  }
  void drain() {
    if (s_in) (void)hipStreamSynchronize(s_in);
    if (s_run) (void)hipStreamSynchronize(s_run);
    if (s_out) (void)hipStreamSynchronize(s_out);
    (void)hipGetLastError();
  }
  ~HostPipe() {
    drain();
    for (int i = 0; i < SLOTS; ++i) {
      if (dX[i]) (void)hipFree(dX[i]);
      if (dY[i]) (void)hipFree(dY[i]);
      if (hIn[i]) (void)hipHostFree(hIn[i]);
      if (hOut[i]) (void)hipHostFree(hOut[i]);
      if (landed[i]) (void)hipEventDestroy(landed[i]);
      if (applied[i]) (void)hipEventDestroy(applied[i]);
      if (returned[i]) (void)hipEventDestroy(returned[i]);
    }
    if (s_in) (void)hipStreamDestroy(s_in);
    if (s_run) (void)hipStreamDestroy(s_run);
    if (s_out) (void)hipStreamDestroy(s_out);
  }
};

void host_pipe_destroy(HostPipe* pipe) { delete pipe; }

namespace {

#define CARD_HIP(call)                                  \
  do {                                                  \
    const hipError_t e__ = (call);                      \
    if (e__ != hipSuccess) return pipe.hip_fail(e__);   \
  } while (0)

// UP + RUN + DOWN of one block; shared by the staged and the caller-pinned form, which differ only in where the two host
// ends of the copies are.  Device-side slot reuse is ordered by events alone (no host synchronisation):
//   UP(b) waits for RUN(b - SLOTS) (it read dX[slot]);  RUN(b) waits for UP(b) and for DOWN(b - SLOTS) (it read dY[slot]).
struct BlockProgram {
  sri_apply_t* h;
  HostPipe& pipe;
  size_t rows;

  template <class Up, class Down>
  int block(int slot, size_t cols, size_t base_off, Up up, Down down) {
    if (pipe.used[slot]) CARD_HIP(hipStreamWaitEvent(pipe.s_in, pipe.applied[slot], 0));
    CARD_HIP(up(pipe.dX[slot], pipe.s_in));
    CARD_HIP(hipEventRecord(pipe.landed[slot], pipe.s_in));
    CARD_HIP(hipStreamWaitEvent(pipe.s_run, pipe.landed[slot], 0));
    if (pipe.used[slot]) CARD_HIP(hipStreamWaitEvent(pipe.s_run, pipe.returned[slot], 0));
    const double* d_base = base_off != 0 ? reinterpret_cast<const double*>(pipe.dX[slot] + base_off) : nullptr;
    const int rc = apply_device_panel(h, reinterpret_cast<const double*>(pipe.dX[slot]), rows, reinterpret_cast<double*>(pipe.dY[slot]), rows,
                                      cols, d_base, pipe.s_run);
    if (rc != SRI_OK) return rc;
    CARD_HIP(hipEventRecord(pipe.applied[slot], pipe.s_run));
    CARD_HIP(hipStreamWaitEvent(pipe.s_out, pipe.applied[slot], 0));
    CARD_HIP(down(pipe.dY[slot], pipe.s_out));
    CARD_HIP(hipEventRecord(pipe.returned[slot], pipe.s_out));
    pipe.used[slot] = true;
    return SRI_OK;
  }
};

// The card as run_host_pipeline() sees it: pinned staging slots on the host side of every copy.
struct StagedCard {
  static constexpr int SLOTS = HostPipe::SLOTS;
  BlockProgram prog;
  HostPipe& pipe;
  char* stage_in(int slot) { return pipe.hIn[slot]; }
  char* stage_out(int slot) { return pipe.hOut[slot]; }
  int enqueue(int slot, size_t /*c0*/, size_t cols, size_t in_bytes, size_t out_bytes, size_t base_off) {
    return prog.block(
        slot, cols, base_off,
        [&](char* dev, hipStream_t s) { return hipMemcpyAsync(dev, pipe.hIn[slot], in_bytes, hipMemcpyHostToDevice, s); },
        [&](char* dev, hipStream_t s) { return hipMemcpyAsync(pipe.hOut[slot], dev, out_bytes, hipMemcpyDeviceToHost, s); });
  }
  int wait_uploaded(int slot) {
    CARD_HIP(hipEventSynchronize(pipe.landed[slot]));
    return SRI_OK;
  }
  int wait_downloaded(int slot) {
    CARD_HIP(hipEventSynchronize(pipe.returned[slot]));
    return SRI_OK;
  }
  void drain() { pipe.drain(); }
};

bool is_pinned_host(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
// EVERY byte of [p, p + bytes) page-locked: the allocations (hipHostMalloc) or registrations (hipHostRegister) that the runtime
// knows are walked from p on — each must be host memory and must start where the previous one ended (the pointer attributes
// RANGE_START_ADDR / RANGE_SIZE give the allocation or registration an address belongs to).  Checking the two end bytes only (round 4) accepted a range that
// spans two pinned buffers with a pageable gap between them, and a registration shorter than the range.
bool is_pinned_host_range(const void* p, size_t bytes) {
  const char* at = static_cast<const char*>(p);
  const char* const end = at + bytes;
  while (at < end) {
    if (!is_pinned_host(at)) return false;
    // (hipMemGetAddressRange reports the size but a NULL base for memory registered with hipHostRegister — measured on ROCm 7.2;
    //  the pointer attributes give both for registered and for hipHostMalloc memory)
    void* base = nullptr;
    size_t size = 0;
    if (hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, const_cast<char*>(at)) != hipSuccess ||
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, const_cast<char*>(at)) != hipSuccess || base == nullptr || size == 0) {
      (void)hipGetLastError();
      return false;
    }
    const char* const next = static_cast<const char*>(base) + size;
    if (next <= at) return false;
    at = next;
  }
  return true;
}

// The caller declared X, Y and x_base page-locked (hipHostMalloc, or its own hipHostRegister held across the call): the same
// block program with the caller's memory at the host end of every copy; nothing is staged and no CPU thread copies.
int run_caller_pinned(sri_apply_t* h, HostPipe& pipe, const double* X, size_t ldx, const double* x_base, double* Y, size_t ldy, size_t M,
                      size_t cb) {
  const size_t rows = h->rows, col_bytes = rows * sizeof(double);
  if (!is_pinned_host_range(X, ((M - 1) * ldx + rows) * sizeof(double)) || !is_pinned_host_range(Y, ((M - 1) * ldy + rows) * sizeof(double)) ||
      (x_base != nullptr && !is_pinned_host_range(x_base, M * sizeof(double))))
    return SRI_EINVAL;                                            // the promise does not hold: refuse rather than fault
  BlockProgram prog{h, pipe, rows};
  size_t b = 0;
  for (size_t c0 = 0; c0 < M; c0 += cb, ++b) {
    const int slot = static_cast<int>(b % HostPipe::SLOTS);
    const size_t cols = std::min(cb, M - c0);
    const size_t base_off = x_base != nullptr ? slot_base_offset(rows, cols) : 0;
    const int rc = prog.block(
        slot, cols, base_off,
        [&](char* dev, hipStream_t s) {
          hipError_t e = ldx == rows ? hipMemcpyAsync(dev, X + c0 * ldx, cols * col_bytes, hipMemcpyHostToDevice, s)
                                     : hipMemcpy2DAsync(dev, col_bytes, X + c0 * ldx, ldx * sizeof(double), col_bytes, cols, hipMemcpyHostToDevice, s);
          if (e == hipSuccess && x_base != nullptr) e = hipMemcpyAsync(dev + base_off, x_base + c0, cols * sizeof(double), hipMemcpyHostToDevice, s);
          return e;
        },
        [&](char* dev, hipStream_t s) {
          return ldy == rows ? hipMemcpyAsync(Y + c0 * ldy, dev, cols * col_bytes, hipMemcpyDeviceToHost, s)
                             : hipMemcpy2DAsync(Y + c0 * ldy, ldy * sizeof(double), dev, col_bytes, col_bytes, cols, hipMemcpyDeviceToHost, s);
        });
    if (rc != SRI_OK) {
      pipe.drain();
      return rc;
    }
  }
  hipError_t e;
  if ((e = hipStreamSynchronize(pipe.s_out)) != hipSuccess || (e = hipStreamSynchronize(pipe.s_run)) != hipSuccess ||
      (e = hipStreamSynchronize(pipe.s_in)) != hipSuccess) {
    pipe.drain();
    return pipe.hip_fail(e);
  }
  return SRI_OK;
}
#undef CARD_HIP

int run_host(sri_apply_t* h, const double* X, size_t ldx, const double* x_base, double* Y, size_t ldy, size_t M) {
  if (h == nullptr || (M > 0 && (X == nullptr || Y == nullptr)) || ldx < h->rows || ldy < h->rows) return SRI_EINVAL;
  if (M == 0) return SRI_OK;
  PanelDeviceGuard guard;
  const hipError_t ge = guard.enter(h->device);
  if (ge != hipSuccess) {
    record_hip_error(static_cast<int>(ge));
    (void)hipGetLastError();
    return SRI_EHIP;
  }
  const size_t rows = h->rows;
  size_t cb = host_block_columns(rows, M);
  if (h->host_block_cols != 0) cb = h->host_block_cols;          // SRI_OPT_HOST_BLOCK_COLUMNS (tools/host_panel_bench.py --block-cols): whole tiles
  if (h->pipe == nullptr) {
    const unsigned threads = h->host_copy_threads >= 0 ? static_cast<unsigned>(h->host_copy_threads) : default_copy_threads();
    h->pipe = new (std::nothrow) HostPipe(threads);
  }
  if (h->pipe == nullptr) return SRI_ENOMEM;
  HostPipe& pipe = *h->pipe;
  const bool staged = !h->host_memory_pinned;
  int rc = pipe.reserve(slot_bytes(rows, cb, x_base != nullptr), rows * cb * sizeof(double), staged);
  if (rc != SRI_OK) return rc;
  for (bool& u : pipe.used) u = false;                           // the previous call drained: no event of it orders anything
  if (!staged) return run_caller_pinned(h, pipe, X, ldx, x_base, Y, ldy, M, cb);
  StagedCard card{BlockProgram{h, pipe, rows}, pipe};
  rc = run_host_pipeline(card, pipe.crew, X, ldx, x_base, Y, ldy, rows, M, cb);
  if (rc != SRI_OK) return rc;
  // every DOWN has been waited for by the driver (wait_downloaded of each block); UP and RUN precede it on the card
  return SRI_OK;
}

}  // namespace

}  // namespace sri

extern "C" {

int sri_apply_host(sri_apply_t* h, const double* X, size_t ldx, double* Y, size_t ldy, size_t M) {
  return sri::run_host(h, X, ldx, nullptr, Y, ldy, M);
}

int sri_integrate_host(sri_apply_t* h, const double* B, size_t ldb, const double* x_base, double* Y, size_t ldy, size_t M) {
  if (h == nullptr || h->which != SRI_OP_DNN_INV || x_base == nullptr) return SRI_EINVAL;
  return sri::run_host(h, B, ldb, x_base, Y, ldy, M);
}

int sri_apply_set_host_copy_threads(sri_apply_t* h, int threads) {
  if (h == nullptr || threads < -1 || threads > 64) return SRI_EINVAL;
  if (h->pipe != nullptr && h->host_copy_threads != threads) {   // the crew is built with the pipe: rebuild on the next call
    sri::PanelDeviceGuard guard;
    (void)guard.enter(h->device);
    sri::host_pipe_destroy(h->pipe);
    h->pipe = nullptr;
  }
  h->host_copy_threads = threads;
  return SRI_OK;
}

int sri_apply_host_trim(sri_apply_t* h) {
  if (h == nullptr) return SRI_EINVAL;
  if (h->pipe != nullptr) {
    sri::PanelDeviceGuard guard;
    (void)guard.enter(h->device);
    h->pipe->trim();
  }
  return SRI_OK;
}

int sri_apply_set_host_memory_pinned(sri_apply_t* h, int pinned) {
  if (h == nullptr) return SRI_EINVAL;
  h->host_memory_pinned = pinned != 0;
  return SRI_OK;
}

}  // extern "C"
