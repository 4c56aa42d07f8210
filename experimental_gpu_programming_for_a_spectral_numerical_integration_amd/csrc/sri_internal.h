// sri_internal.h — declarations shared by the translation units of libsri_hip.so (product code).
#ifndef SRI_INTERNAL_H
#define SRI_INTERNAL_H

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "sri_capi.h"

namespace sri {

// ---- host formers (sri_host.cpp) -------------------------------------------------------------
int cheb_points(uint32_t n, double L, double* x);
int cheb_coeffs(uint32_t n, double* c);
int cheb_dn(uint32_t n, double* D, size_t ld);
int cheb_blocks(uint32_t n, const double* D, size_t ld, double* Dnn, size_t ldnn, double* Din);
int dense_inverse(uint32_t m, const double* A, size_t lda, double* Ainv, size_t ldi);
int cheb_integration(uint32_t n, double* Dnn_inv, size_t ld, double* Din);
int ivp(uint32_t m, uint32_t s, const double* Din, const double* x_base, double* out, size_t ld);
int cheb_interp(uint32_t n_from, uint32_t n_to, double* P, size_t ld);
// the fine integration operator (and its initial-value weights) as a coarser grid sees it — sri_host.cpp; may throw bad_alloc
int coarse_view_of_fine_operator(const double* A_f, size_t lda, const double* w_f, uint32_t n_f, uint32_t n_c, std::vector<double>& A_c,
                                 std::vector<double>& w_c);

// ---- the library's ONE environment hook (sri_host.cpp) ------------------------------------------------------------------
// SRI_DEBUG="key=value,key=value": A/B scaffolding for tools/ and tests/ (force_geo=m|f, strip=0, parts_ext=0, valu_prolong=1,
// comm_force_ragged=1), read where a handle is CREATED.  Everything a caller may want to switch is a handle option
// (sri_apply_set_option); nothing else in the library reads the environment.
std::string debug_option(const char* key);   // "" when SRI_DEBUG is unset or does not name `key`

// ---- error bookkeeping (sri_apply.hip): what sri_last_hip_error() reports -----------------------
void record_hip_error(int hip_error);

// ---- fused Picard iteration of the quaternion system (sri_apply.hip, used by sri_rod.hip) ----------
// G_next = 1/2 A(K) (Dn_NN^-1 G - (Dn_NN^-1 Dn_IN) x_base^T),  K = Phi(X_row) qe, in ONE launch of the integration
// apply: the right-hand side of main.cpp:71-84 is formed in the store stage from the accumulators.  All pointers are
// device pointers; leg1 holds P1 of utilities.h:54-59 (= 2X - 1) per row, zero padded to padded_rows(h).
struct QuatEpilogue {
  const double* qe;     // 9 per rod
  const double* leg1;
  // optional convergence monitor: the launch writes partial[0 .. *partials) (one value per workgroup and wave column:
  // the largest |tip - tips_prev| it saw, NaN if any); the caller takes the maximum
  const double* tips_prev = nullptr;
  double* partial = nullptr;
  // optional: device word; the launch is a no-op when it is non-zero (lets a fixed-point loop be enqueued ahead of its
  // own convergence test)
  const unsigned long long* stop = nullptr;
  // optional: loop control inside the launch — before it does anything, the launch folds the `fold_n` partial maxima of the
  // PREVIOUS step (`fold`), and if their maximum is <= fold_tol (converged: state[0] = 1) or `iteration` completed steps
  // have reached fold_max_iter (state[0] = 2) it records {stop, iteration, bits(maximum)} in `state` and returns
  const double* fold = nullptr;
  uint32_t fold_n = 0;
  uint32_t iteration = 0;
  double fold_tol = 0.0;
  unsigned long long fold_max_iter = 0;
  unsigned long long* state = nullptr;
};
// `partials`: in = capacity of quat.partial (entries), out = entries written by this launch.
int integrate_quat_step(sri_apply_t* h, const double* dG, size_t ldg, const double* d_x_base, const QuatEpilogue& quat,
                        double* dG_next, size_t ldn, size_t M, double* d_tips, void* stream, uint32_t* partials = nullptr);
// a rows_out x kdim operator with rank-1 weights (kdim >= 16, kdim % 16 != 0), usable with integrate_quat_step only — sri_apply.hip
int create_interpolation_operator(sri_apply_t** out, uint32_t rows_out, uint32_t kdim, const double* A, size_t lda, const double* w, int device);
// The CLOSING product of a rod solve (sri_rod.hip): Q = Dn_NN^-1 G - w x_base^T as sri_integrate forms it, and from the same
// accumulators the position right-hand side b = R(q) e1 of main.cpp:121-140 (dB: rows x 3 M / 4, ldb > rows when rows is odd) — one
// launch (EPI_QB), the bits of sri_integrate followed by position_rhs_kernel.  M a multiple of 4.  SRI_EUNSUPPORTED where the handle
// has no such variant (row counts that are multiples of 16, operators outside the 256-row tile geometry): the caller composes the two.
// With `crit` (EPI_QBC; the 128- and 256-row geometries, paired k order, at least three k chunks, ldg even and >= rows + rows % 2, dG
// 16-byte aligned — else SRI_EUNSUPPORTED; row counts that are multiples of 16 take part only in this form, which then applies the
// rank-1 term in its store stage): the same launch also forms d = 1/2 A(K) Q - G in registers (K = Phi(X_row) qe as integrate_quat_step does)
// and leaves, by atomic max into words the caller has zeroed, words[0] = bits of max |d| and words[1 + B] = bits of the largest sum of
// d^2 that any column has over rows [64 B, 64 B + 64) (criterion_wave_rows(h) != 0: the handle has such a variant; rows <= 4096).
// `clear` (optional): kCriterionWords words that the launch zeroes for the caller's NEXT launch.
constexpr uint32_t kCriterionWords = 65;   // max |d| + one word per 64-row segment of operators of up to 4096 rows
struct FixedPointCriterion {
  const double* qe;     // 9 per rod
  const double* leg1;   // P1 per (padded) row, as QuatEpilogue's
  unsigned long long* words;
  unsigned long long* clear = nullptr;
};
int integrate_with_position_rhs(sri_apply_t* h, const double* dG, size_t ldg, const double* d_x_base, double* dQ, size_t ldq, double* dB,
                                size_t ldb, size_t M, void* stream, const FixedPointCriterion* crit = nullptr);
uint32_t criterion_wave_rows(const sri_apply_t* h);   // wave rows per row block of the geometry EPI_QBC runs in (2: 128-row tiles, 4: 256-row tiles; 0: no such variant)
uint32_t max_partials(const sri_apply_t* h);   // upper bound of *partials for any launch of this handle
uint32_t padded_rows(const sri_apply_t* h);
int ivp_weights(const sri_apply_t* h, double* w_host);   // w = Dn_NN^-1 Dn_IN as the handle applies it (rows doubles); `device` current

// ---- the latency kernel for 64-row operators and small batches: one tile per workgroup (sri_apply_small.hip) ---------
constexpr uint32_t kSmallRows = 64;
bool small_supported(uint32_t rows);
bool small_batch(size_t M, int num_cus, uint32_t max_wgs);                // M columns fit one resident round of tiles
void small_pack(const double* A, size_t lda, std::vector<double>& out);   // rows x rows column-major -> fragment order
int small_prepare(int dev);                                               // `dev` must be current
int small_launch(const double* d_apack, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, double* d_tips, int dev,
                 int num_cus, uint32_t max_wgs, void* stream);

// ---- on-chip rod pipeline for small grids (sri_rod_resident.hip, used by sri_rod.hip) -----------------------------
struct ResidentRod;
bool resident_supported(uint32_t m);                 // 2 <= m = N - 1 <= 63
// `device` must be current.  A_fine / w_fine / n_fine: this handle is the coarse start of an n_fine-node grid whose integration
// operator is A_fine ((n_fine-1)^2, column-major) with initial-value weights w_fine — see coarse_view_of_fine_operator.
int resident_create(uint32_t n, int device, int num_cus, ResidentRod** out, const double* A_fine = nullptr,
                    const double* w_fine = nullptr, uint32_t n_fine = 0);
int resident_destroy(ResidentRod* h);
int resident_integrate(ResidentRod* h, const double* d_qe, size_t rods, const double* q_init, const double* r_init, double* dQ,
                       size_t ldq, double* dR, size_t ldr, double tol, uint32_t max_iter, uint32_t* iters_out,
                       double* last_update_out, void* stream, bool coarse_start = true,    // coarse_start: grids of > 16 nodes begin
                       bool defer_stats = false);                                           // from an on-chip 16-node solution;
                                                                                            // defer_stats: do not wait (resident_collect)
int resident_collect(ResidentRod* h, uint32_t* iters_out, double* last_update_out, void* stream);

// ---- fragment geometry shared by the host packer and the kernel (sri_apply.hip) ---------------
// One v_mfma_f64_16x16x4_f64 computes D[i][j] += sum_k P[i][k] Q[k][j] (i, j < 16, k < 4); lane l feeds
// P[l & 15][l >> 4] and Q[l >> 4][l & 15] and receives D[(l >> 4) + 4*reg][l & 15], reg = 0..3.
// The kernel feeds the X fragment as P and the operator fragment as Q, i.e. it accumulates the
// TRANSPOSED tile: i = column of Y, j = row of Y.  The 16 consecutive lanes of a quarter-wave then
// hold 16 different ROWS of one column, and by interleaving the rows of two vertically adjacent
// tiles (tm = 2t, 2t+1) every lane owns two ADJACENT rows: a Y store is 16 bytes per lane and
// 256 contiguous bytes (two full cache lines) per quarter-wave, which is how the memory pipeline
// consumes a wave's store.
//   row slot j of tile tm (tm counted inside the wave's 64-row block)  <->  row 32*(tm>>1) + 2*j + (tm&1)
constexpr int kChunkK = 16;   // k extent of one pipeline stage (BK)

inline constexpr int row_of_slot(int tm, int j) { return 32 * (tm >> 1) + 2 * j + (tm & 1); }
// k index inside a 16-k chunk consumed by MFMA (k pair j, element e) from the lanes' k slot g = lane >> 4.
//   paired: a lane's 16-byte X read holds k = 2p, 2p+1, so an instruction sums four k of equal parity: 0,2,4,6 | 1,3,5,7 |
//     8,10,... — the fewest LDS instructions; harmless for the integration operators (||Dn_NN^-1||_inf = 1) and for
//     differentiation on grids up to N = 512 (within 2e-13 of the CPU reference on smooth fields at N = 256).
//   ascending: instruction t = 2j + e sums k = 4t .. 4t+3, so every output element receives its products in the
//     reference's order k = 0, 1, 2, ... (main.cpp:172) — the only arithmetic difference left is the fused multiply-add.
//     Costs 8-byte X-fragment reads (one per MFMA).  Used for differentiation operators of more than 512 rows, where the
//     paired order's partial sums, 4 terms out of balance under alternating signs, cost accuracy (sri_apply.hip).
inline constexpr int k_of_ascending(int j, int g, int e) { return 8 * j + 4 * e + g; }
inline constexpr int k_of_paired(int j, int g, int e) { return 8 * j + 2 * g + e; }

// ---- host panels (sri_host_panels.hip): the pinned staging ring + copy threads behind sri_apply_host / sri_integrate_host ----
struct HostPipe;
void host_pipe_destroy(HostPipe* pipe);              // the handle's device must be current
// the apply on a block that already sits on the device (sri_apply.hip); d_x_base != nullptr: sri_integrate's form
int apply_device_panel(sri_apply_t* h, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, const double* d_x_base, void* stream);

#if defined(SRI_DEVICE_HELPERS)   /* defined by the .hip translation units after <hip/hip_runtime.h> */
}  // namespace sri
#include <atomic>
#include <string>
#include <tuple>
#include <utility>

// The apply handle (sri_apply.hip creates, launches and destroys it; sri_host_panels.hip hangs its pipe on it).
struct sri_apply_handle {
  int device = 0;
  uint32_t rows = 0;             // columns of the operator = rows of the panels it is applied to (k)
  uint32_t rows_out = 0;         // rows of the operator = rows of the result (== rows for every operator of the C ABI)
  int which = -1;
  int geo = 0;
  uint32_t bm = 0, nk = 0, row_blocks = 0;
  int num_cus = 0;
  uint32_t max_workgroups = 0;   // 0 = fill the chip; else cap of the persistent grid (leave CUs to a collective)
  double* d_apack = nullptr;
  double* d_apack_small = nullptr;   // 64-row operators: fragment order of sri_apply_small.hip (small plain applies take that kernel)
  double* d_w = nullptr;        // Dnn_inv * Din (padded), for sri_integrate
  bool ascending = false;       // k order of this handle's launches (see kAscendingAboveRows)
  // "Parts" (sri_apply.hip): for plain and rank-1-epilogue launches the operator may be cut into a main part and a last part
  // of its own geometry (rows [row0, row0 + rows), all k), each packed and launched separately on the caller's stream.
  // num_parts == 0: one geometry for the whole operator (always what the fused step and sri_apply_many use).
  struct Part {
    int geo = 0;
    uint32_t row0 = 0, rows = 0, row_blocks = 0;
    double* d_apack = nullptr;
  };
  Part parts[2];
  int num_parts = 0;
  bool parts_have_wide = false;
  std::string parts_label;           // what sri_apply_last_kernel reports for a launch in parts
  const char* parts_name = "none";
  std::string parts_label_lean;
  const char* parts_name_lean = "none";
  // Alternates (sri_column_plan.h): the WHOLE operator packed once more for each finer tile geometry; run() gives them the
  // columns behind the primary path's full rounds of tiles, or a small panel as a whole.  Same k order, same bits.
  struct Alternate {
    int geo = -1;
    uint32_t row_blocks = 0;
    double* d_apack = nullptr;
  };
  Alternate alternates[2];
  bool crit_on_mid = true;               // EPI_QBC of a 256-row-geometry operator runs on its 128-row alternate (sri_apply.hip: criterion_alternate)
  int num_alternates = 0;
  bool column_plan = true;           // sri_apply_set_column_plan / SRI_OPT_COLUMN_PLAN
  std::string mixed_labels[2][2];    // [primary lean][alternate]: what sri_apply_last_kernel reports for a primary + alternate launch
  bool lean_tail = true;             // odd-row-count operators: take the lean kernels where run() finds them safe (sri_apply_set_lean_tail)
  bool use_parts = true;             // SRI_OPT_ROW_PARTS: 0 launches the operator whole in its own geometry although it is kept in parts
  bool use_small = true;             // SRI_OPT_SMALL_KERNEL: 0 keeps 64-row operators on the general kernel
  bool xcd_walk = true;              // SRI_OPT_XCD_TILE_WALK: 0 keeps the plain tile walk for operators of several row blocks
  size_t host_block_cols = 0;        // SRI_OPT_HOST_BLOCK_COLUMNS: columns per block of the host-panel pipeline (0 = automatic)
  // SRI_OPT_PLAN_CALIBRATION: where the column plan's cost model wants columns in a finer geometry, the first call at that
  // column count times both schedules on the call's own panel (same bits either way) and the handle remembers the winner
  bool plan_calibrate = true;
  struct PlanVerdict {
    size_t cols = 0;
    bool rank1 = false, use_plan = false;
    float own_us = 0.0f, plan_us = 0.0f;
    uint32_t calls = 0;                    // calls at this column count since the first measurement; at kPlanRecheckCall the schedules are
                                           // timed once more — on a device that is warm by then — and that verdict stays
  };
  static constexpr uint32_t kPlanRecheckCall = 16;
  static constexpr int kPlanVerdicts = 16;
  PlanVerdict plan_verdicts[kPlanVerdicts];
  int plan_verdict_count = 0, plan_verdict_next = 0;
  hipEvent_t plan_events[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::atomic<const char*> last_kernel{"none"};   // diagnostics only; written by every launch (atomic: no torn read if a second thread asks)
  sri::HostPipe* pipe = nullptr;   // sri_apply_host / sri_integrate_host: streams, events, staging and device slots, built on first use
  int host_copy_threads = -1;      // CPU threads that move a host panel into / out of the pinned staging slots; -1 = default
  bool host_memory_pinned = false; // the caller promised page-locked X / Y / x_base: no staging (sri_apply_set_host_memory_pinned)
};

namespace sri {
// Enqueue `kernel` and return the launch's OWN status.  (hipLaunchKernelGGL + hipGetLastError reports whatever failed
// last on this thread — also a call of the application's, or a refused hipHostRegister — as a failed launch.)
template <class... Params, class... Args, size_t... I>
inline hipError_t launch_kernel_impl(void (*kernel)(Params...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                                     std::index_sequence<I...>, Args&&... args) {
  std::tuple<Params...> values{static_cast<Params>(std::forward<Args>(args))...};
  void* ptrs[] = {static_cast<void*>(&std::get<I>(values))...};
  return hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, stream);
}
template <class... Params, class... Args>
inline hipError_t launch_kernel(void (*kernel)(Params...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args&&... args) {
  static_assert(sizeof...(Params) == sizeof...(Args), "one argument per kernel parameter");
  return launch_kernel_impl(kernel, grid, block, lds, stream, std::index_sequence_for<Params...>{}, std::forward<Args>(args)...);
}

// G = 1/2 A(K) q of main.cpp:71-84 with hk = K/2:  A(K) = [0 -k0 -k1 -k2; k0 0 k2 -k1; k1 -k2 0 k0; k2 k1 -k0 0].
// One product and two fused multiply-adds per component — f64 VALU work shares the MFMA pipe.  The stand-alone
// kernel of sri_rod.hip and the fused epilogue of sri_apply.hip use the same
// function, so fused and unfused iterations agree bit for bit.
__device__ __forceinline__ void quat_rhs_half(double h0, double h1, double h2, double w, double x, double y, double z,
                                              double (&g)[4]) {
  g[0] = __builtin_fma(-h2, z, __builtin_fma(-h1, y, -(h0 * x)));
  g[1] = __builtin_fma(-h1, z, __builtin_fma(h2, y, h0 * w));
  g[2] = __builtin_fma(h0, z, __builtin_fma(-h2, x, h1 * w));
  g[3] = __builtin_fma(-h0, y, __builtin_fma(h1, x, h2 * w));
}

#endif

}  // namespace sri

#endif  // SRI_INTERNAL_H
