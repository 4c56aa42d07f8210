/*
 * sri_capi.h — C ABI of libsri_hip.so, the MI355X-native spectral-integration hot path.
 *
 * The reference (aGotelli/experimental_gpu_programming_for_a_spectral_numerical_integration)
 * has no plugin / FFI boundary: its API is three header-only templates in
 * include/chebyshev_differentiation.h called straight from main.cpp.  This header is the
 * boundary a maintainer would bind instead; include/chebyshev_differentiation.h in THIS repo
 * re-exposes the reference's three template signatures on top of it.
 *
 * Conventions
 *   - every matrix is COLUMN-MAJOR fp64 with an explicit leading dimension (Eigen's default
 *     storage, so `M.data()`, `M.outerStride()` can be passed as is);
 *   - a "state panel" X is rows x M: one column per (rod, state component), rods concatenated
 *     along columns (main.cpp:113,130-133: Q_stack[c*(N-1)+i] == column-major (N-1) x 4);
 *   - all entry points return 0 (SRI_OK) or a negative sri_status; nothing throws across the ABI;
 *   - host formers are re-entrant; one sri_apply_t per device, not to be driven from two threads
 *     at once without external locking;
 *   - sri_apply() takes DEVICE pointers and only enqueues work on the given HIP stream
 *     (no allocation, no synchronisation: it is graph-capturable).
 *
 * file:line citations are relative to the reference tree.
 */
#ifndef SRI_CAPI_H
#define SRI_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRI_ABI_VERSION 5

typedef enum sri_status {
    SRI_OK = 0,
    SRI_EINVAL = -1,       /* null pointer, n < 2, leading dimension too small ...            */
    SRI_ENOMEM = -2,       /* host or device allocation failed                                */
    SRI_ESINGULAR = -3,    /* zero pivot while inverting Dn_NN                                 */
    SRI_EHIP = -4,         /* a HIP runtime call failed (sri_last_hip_error() has the code)   */
    SRI_ENODEVICE = -5,    /* no gfx950 device / bad device ordinal                           */
    SRI_EUNSUPPORTED = -6, /* valid request this build cannot serve                           */
    SRI_ENOTCONVERGED = -7,/* sri_rod_integrate_quaternions: update still above tol at max_iter */
    SRI_ECOMM = -8         /* an RCCL call failed (sri_last_rccl_error() has the ncclResult_t)   */
} sri_status;

/* Which constant left-hand matrix an apply handle holds. */
typedef enum sri_operator {
    SRI_OP_DN = 0,       /* D_N, N x N                    (chebyshev_differentiation.h:59-108; used as
                            I_s (x) D_N at main.cpp:97)                                          */
    SRI_OP_DNN_INV = 1,  /* Dn_NN^-1, (N-1) x (N-1)       (main.cpp:158-159; applied at main.cpp:172) */
    SRI_OP_DNN = 2       /* Dn_NN,    (N-1) x (N-1)       (main.cpp:94, 158)                          */
} sri_operator;

/* ---------------------------------------------------------------------------------------- */
/* Host formers (O(N^2) / O(N^3) once per N; IEEE fp64, libm cos on the host so that D_N is   */
/* bit-identical to what the reference forms — SURVEY.md §7 H3).                              */
/* ---------------------------------------------------------------------------------------- */

/* ComputeChebyshevPoints<N, L>()  chebyshev_differentiation.h:19-30.
 * x[j] = (L/2)(1 + cos(pi j/(n-1))), j = 0..n-1, descending from L to 0.  x: n doubles. */
int sri_cheb_points(uint32_t n, double L, double *x);

/* GetCoefficients_c<N>()  chebyshev_differentiation.h:37-52.
 * c[i] = (-1)^i * (2 at both ends, 1 elsewhere).  c: n doubles. */
int sri_cheb_coeffs(uint32_t n, double *c);

/* getDn<N>()  chebyshev_differentiation.h:59-108.
 * D: n x n column-major, leading dimension ld >= n. */
int sri_cheb_dn(uint32_t n, double *D, size_t ld);

/* Block slicing  main.cpp:94-95, 158, 160.
 * Dnn = D[0:n-1, 0:n-1] (ldnn >= n-1), Din = D[0:n-1, n-1] (n-1 doubles).  Either output may be
 * NULL. */
int sri_cheb_blocks(uint32_t n, const double *D, size_t ld, double *Dnn, size_t ldnn, double *Din);

/* Spectral integration matrix  main.cpp:157-160:  Dnn_inv = (getDn<n>()[0:n-1,0:n-1])^-1 by
 * partial-pivot LU (what Eigen's .inverse() does for sizes > 4), plus the base-node influence
 * column Din (may be NULL).  Dnn_inv: (n-1) x (n-1), ld >= n-1. */
int sri_cheb_integration(uint32_t n, double *Dnn_inv, size_t ld, double *Din);

/* Initial-value term  main.cpp:162-164 (positions), main.cpp:100,109 (quaternions):
 * ivp(i,c) = Din[i] * x_base[c];  ivp is m x s column-major, ld >= m. */
int sri_ivp(uint32_t m, uint32_t s, const double *Din, const double *x_base, double *ivp, size_t ld);

/* Not in the reference: barycentric interpolation matrix between two of its Chebyshev grids on [0, 1]
 * (nodes of chebyshev_differentiation.h:26), P: n_to x n_from column-major, ld >= n_to; values at the n_from nodes ->
 * values at the n_to nodes.  sri_rod_* uses it to start the fixed point of a fine grid from a 32-node solution. */
int sri_cheb_interp(uint32_t n_from, uint32_t n_to, double *P, size_t ld);

/* Not in the reference: the integration operator of a FINE grid (A_fine = Dn_NN^-1 of n_fine nodes, (n_fine-1)^2 column-major,
 * lda >= n_fine - 1; w_fine = Dn_NN^-1 Dn_IN) as a coarser grid of n_coarse < n_fine nodes sees it:
 *   A_coarse = R_u A_fine P~   ((n_coarse-1)^2, ldc >= n_coarse - 1),     w_coarse = -R [-w_fine ; 1]
 * with R the fine->coarse interpolation of sri_cheb_interp (base node last), R_u its columns of unknown fine nodes and P~ the
 * interpolation from the unknown coarse nodes to the unknown fine nodes.  In exact arithmetic this IS the coarse grid's own
 * operator; as computed it carries the fine grid's rounding-level deviation instead of the coarse grid's, which is what a
 * coarse solve needs when it only starts the fixed point of the fine grid (sri_rod_*: one fine sweep instead of six). */
int sri_cheb_coarse_view(uint32_t n_fine, const double *A_fine, size_t lda, const double *w_fine, uint32_t n_coarse,
                         double *A_coarse, size_t ldc, double *w_coarse);

/* ---------------------------------------------------------------------------------------- */
/* The apply:  Y[rows x M] = A[rows x rows] * X[rows x M]   (main.cpp:172; Kronecker form      */
/* main.cpp:97).  A is formed on the host once, packed into MFMA-fragment order and kept on    */
/* the device by the handle; X and Y are device buffers owned by the caller.                  */
/* ---------------------------------------------------------------------------------------- */
typedef struct sri_apply_handle sri_apply_t;

/* Form operator `which` for n Chebyshev nodes and upload it to HIP device `device`. */
int sri_apply_create(sri_apply_t **out, uint32_t n, int which, int device);

/* Same, for a caller-supplied square matrix A (rows x rows, column-major, lda >= rows). */
int sri_apply_create_matrix(sri_apply_t **out, uint32_t rows, const double *A, size_t lda, int device);

/* Cap the number of persistent workgroups a launch uses (0 = fill the chip, the default).  The apply
 * kernel occupies every CU for the whole launch; a caller that overlaps a collective (RCCL) with it
 * leaves a few CUs free this way so that the collective's kernel can run concurrently. */
int sri_apply_set_max_workgroups(sri_apply_t *h, uint32_t max_workgroups);

/* rows (= cols) of the operator held by the handle: n for SRI_OP_DN, n-1 otherwise. */
int sri_apply_dim(const sri_apply_t *h, uint32_t *rows);

/* Enqueue Y = A * X on `hip_stream` (a hipStream_t, NULL = the device's null stream).
 * dX, dY: device pointers, column-major rows x M, ldx, ldy >= rows.  X and Y must not overlap.
 * M may be any value >= 0; no alignment requirement beyond that of a double (panels whose columns start on
 * 128-byte lines additionally get non-temporal stores).  Leading dimensions up to 4 194 000 doubles (128 columns
 * must span less than 4 GiB: SRI_EUNSUPPORTED beyond). */
int sri_apply(sri_apply_t *h, const double *dX, size_t ldx, double *dY, size_t ldy, size_t M,
              void *hip_stream);

/* `count` INDEPENDENT panels in ONE persistent launch: Y_i = A * X_i, i < count, every panel rows x M with the same
 * M, ldx, ldy.  dXs, dYs: DEVICE arrays of `count` device pointers (as in batched BLAS).  A caller whose batches arrive as
 * many small panels — BASELINE config 2 is 4 096 rods: 16.8 MB, a 5 us kernel of which 2.9 us are the launch floor —
 * pays the launch once and streams at the rate of one large panel.  Results are bit-identical to `count` sri_apply calls.
 * Operators of at least 16 rows; plain apply only (no tips, no initial-value term). */
int sri_apply_many(sri_apply_t *h, const double *const *dXs, size_t ldx, double *const *dYs, size_t ldy, size_t M,
                   size_t count, void *hip_stream);

/* sri_apply that additionally writes the tip-node value of every column, d_tips[j] = Y[0][j]
 * (node 0 is the rod tip X = L: chebyshev_differentiation.h:26), M device doubles.  This is the "final
 * integrated state" a multi-GPU run all-gathers; emitted by the kernel's epilogue at no extra traffic. */
int sri_apply_tips(sri_apply_t *h, const double *dX, size_t ldx, double *dY, size_t ldy, size_t M,
                   double *d_tips, void *hip_stream);

/* Position/quaternion integration in one call, main.cpp:172 batched:
 *   Y = Dnn_inv * (B - Din * x_base^T)
 * for a handle created with SRI_OP_DNN_INV.  dB: (n-1) x M, d_x_base: M doubles (the base-node
 * value of every column), all device pointers.  The rank-1 initial-value term -(Dnn_inv*Din) * x_base^T is one more
 * link (k = n - 1) of every output element's chain of fused multiply-adds: inside the product where the last k chunk has
 * a free slot ((n - 1) % 16 != 0), in the kernel's store stage otherwise — the same bits either way. */
int sri_integrate(sri_apply_t *h, const double *dB, size_t ldb, const double *d_x_base,
                  double *dY, size_t ldy, size_t M, void *hip_stream);

/* sri_integrate that also emits the tip-node value of every column (see sri_apply_tips). */
int sri_integrate_tips(sri_apply_t *h, const double *dB, size_t ldb, const double *d_x_base,
                       double *dY, size_t ldy, size_t M, double *d_tips, void *hip_stream);

/* For callers holding HOST panels (what include/chebyshev_differentiation.h calls for the reference's main.cpp:166-172):
 * the panel crosses PCIe in blocks of columns — CPU threads copy block b+1 into a pinned staging slot owned by the handle
 * while the copy engines carry block b in, the kernel applies block b-1 and earlier blocks travel back and are copied
 * out — so a call costs about one direction of the link, not in + apply + out.  The library never page-locks, registers
 * or hands to the GPU runtime any of the caller's memory: X, Y and x_base are only read and written by the CPU.
 * X and Y: column-major rows x M, any leading dimension >= rows; Y may be X (in place), otherwise they must not overlap.
 * Returns when Y is complete.  (PCIe-inclusive; never what bench.py times.) */
int sri_apply_host(sri_apply_t *h, const double *X, size_t ldx, double *Y, size_t ldy, size_t M);

/* Host-buffer form of sri_integrate (x_base: M host doubles). */
int sri_integrate_host(sri_apply_t *h, const double *B, size_t ldb, const double *x_base,
                       double *Y, size_t ldy, size_t M);

/* CPU threads that copy a host panel into / out of the staging slots (besides the calling thread, which helps):
 * -1 = default (min(4, hardware threads)), 0 = the calling thread alone. */
int sri_apply_set_host_copy_threads(sri_apply_t *h, int threads);

/* Resident footprint of the host-panel path: three pinned staging slots in and three out plus the same on the device, each as
 * large as the largest block any call of this handle needed (4-32 MiB per slot: up to ~192 MiB pinned + ~192 MiB device), held
 * until sri_apply_destroy — or until this call, which gives them back (streams and events stay; the next host call allocates
 * what it needs again).  A failed enlargement keeps the smaller buffers the handle had. */
int sri_apply_host_trim(sri_apply_t *h);

/* pinned != 0: the caller PROMISES that every X, Y and x_base it passes to the _host entry points of this handle is
 * page-locked host memory (hipHostMalloc, or the caller's own hipHostRegister held across the call).  The copies then go
 * straight from and to the caller's memory — no staging, no CPU copy; the claim is checked over the WHOLE of each range (the
 * allocations / registrations the runtime knows are walked from the first byte to the last) and a call whose buffers are not
 * page-locked throughout returns SRI_EINVAL.  Default 0. */
int sri_apply_set_host_memory_pinned(sri_apply_t *h, int pinned);

/* Launch policy of operators whose row count is not a multiple of 16 (every integration operator of an N = 2^p grid).  Their
 * kernels fetch the last k chunk of each column where it lies — for the panel's LAST column up to 16 - rows % 16 doubles past
 * the column's end (never used: that part of the on-chip image is overwritten before any arithmetic sees it).  enable != 0
 * (default): where those bytes share their 4-KiB page with the panel's last valid byte — so that reading them cannot fault — the
 * launch takes the kernel variant without the edge handling (1.6 % faster at N = 256); everywhere else, and always with
 * enable == 0, the variant that never reads past the panel.  Same results bit for bit.
 * THE CONTRACT IN ONE SENTENCE: with ld >= 16 * ceil(rows / 16) the library never reads outside the caller's allocation; with
 * ld == rows and rows % 16 != 0 it may READ (never write, never use) up to 120 bytes past the panel's last byte, inside the same
 * 4-KiB page, unless sri_apply_set_lean_tail(h, 0) has been called.  (Memory checkers that track allocations byte-exactly will
 * flag that read; the edge-safe variants cost 1.6 %.) */
int sri_apply_set_lean_tail(sri_apply_t *h, int enable);

/* Launch policy of operators of more than 64 rows: which tile geometry takes which columns of a panel.  Every launch walks
 * whole rounds of tiles over the chip, so a panel of 1.1 rounds of 256 x 128 tiles costs 2, and one of a few thousand columns
 * leaves most of the chip idle for a whole tile time.  enable != 0 (default): per call, the operator's own geometry takes the columns of its full rounds and a finer geometry of the
 * same operator (128 x 128 or 64 x 64 tiles) the rest — or all of a small panel — wherever a cost model built from measured
 * tile times expects at least 5 % to win (csrc/sri_column_plan.h; e.g. N = 256, 4 096 columns: 61.7 -> 13.2 us).  The extra
 * launches go to `stream` one after the other.  enable == 0: always the operator's own geometry.  Same results bit for bit
 * (same k order, same chain of fused multiply-adds per output element).  Plain applies and sri_integrate split a panel this
 * way; the fused step (sri_rod_quaternion_step) takes ONE geometry per launch — its own, or a finer one for ALL columns of a
 * small batch (the number of partial maxima it writes and the name sri_rod_last_apply_kernel reports change with that choice);
 * sri_apply_many and handles with a workgroup cap (sri_apply_set_max_workgroups) keep the operator's own geometry. */
int sri_apply_set_column_plan(sri_apply_t *h, int enable);

/* Every per-handle switch under one entry point (the dedicated setters above remain).  Nothing in the library reads the
 * environment for these: a process that wants another default calls this after sri_apply_create. */
typedef enum sri_option {
    SRI_OPT_COLUMN_PLAN = 1,        /* = sri_apply_set_column_plan (default 1)                                              */
    SRI_OPT_LEAN_TAIL = 2,          /* = sri_apply_set_lean_tail (default 1)                                                */
    SRI_OPT_ROW_PARTS = 3,          /* 0: launch the operator whole, in ONE tile geometry, although it is kept in parts
                                       (rows off the multiples of 64; default 1).  Same bits; A/B timing and tests          */
    SRI_OPT_SMALL_KERNEL = 4,       /* 0: a 64-row operator never takes the one-tile-per-workgroup kernel (default 1)       */
    SRI_OPT_XCD_TILE_WALK = 5,      /* 0: operators of several row blocks walk their tiles row block slowest instead of
                                       grouped by XCD (default 1; same bits, 1.8 x the HBM reads at N = 1024)               */
    SRI_OPT_HOST_COPY_THREADS = 6,  /* = sri_apply_set_host_copy_threads                                                    */
    SRI_OPT_HOST_BLOCK_COLUMNS = 7, /* columns per block of the host-panel pipeline, a multiple of 128; 0 = automatic        */
    SRI_OPT_MAX_WORKGROUPS = 8,     /* = sri_apply_set_max_workgroups                                                       */
    SRI_OPT_PLAN_CALIBRATION = 9    /* 1 (default): the column plan prices its candidates with tile times THIS device
                                       produced (a few launches per geometry, once per handle, on the first planned call's
                                       stream); 0: with the built-in table (csrc/sri_column_plan.h)                         */
} sri_option;
int sri_apply_set_option(sri_apply_t *h, int option, long value);
int sri_apply_get_option(const sri_apply_t *h, int option, long *value);

int sri_apply_destroy(sri_apply_t *h);

/* ---------------------------------------------------------------------------------------- */
/* Batched rod integration: the callers on either side of the apply, device-resident          */
/* (SURVEY.md §8f rows 1-2).  m = n-1 unknown nodes; panels are column-major m x (s*rods):     */
/* per rod 3 columns of K, 4 of Q (w,x,y,z), 3 of b / r.  All pointers are DEVICE pointers.    */
/* ---------------------------------------------------------------------------------------- */
typedef struct sri_rod_handle sri_rod_t;

/* Owns Dn_NN^-1 (an SRI_OP_DNN_INV apply handle), the Chebyshev nodes and a workspace. */
int sri_rod_create(sri_rod_t **out, uint32_t n, int device);
int sri_rod_destroy(sri_rod_t *h);

/* Small grids (3 <= N <= 64) are integrated by an on-chip kernel: one launch
 * runs the whole fixed point (and the position step) with the rod state resident in LDS.  enable = 0 forces the
 * streaming path (one fused launch per iteration) that larger grids use; SRI_EUNSUPPORTED if enabling is impossible. */
int sri_rod_set_resident(sri_rod_t *h, int enable);

/* Grids of more than 16 nodes start the fixed point from a coarse-grid solution: the same rods on a 32-node Chebyshev grid
 * (N > 64: on-chip pipeline, then interpolated by a kernel) or on a 32- / 16-node grid inside the on-chip pipeline itself
 * (32 < N <= 64 / 16 < N <= 32), interpolated to the fine nodes.  The coarse solve uses the FINE grid's operator as the
 * coarse nodes see it, so the start is rounding noise away from the fine fixed point (one fine sweep at N = 64 ... 512).
 * The result is still the fixed point of the fine system, tested and confirmed as without it; only the iteration count
 * changes (*iters counts fine-grid iterations).  enable = 0 starts from Q = q_init at every node instead;
 * SRI_EUNSUPPORTED if enabling is impossible (N <= 16: nothing coarser). */
int sri_rod_set_predictor(sri_rod_t *h, int enable);

/* sri_apply_set_option on every apply handle inside the rod handle (the integration operator, the prolongation) — or one of
 * the rod handle's own switches: */
#define SRI_ROD_OPT_FIRST_SWEEP_CHECK 100 /* 1 (default): after a predicted start, sri_rod_integrate* first runs ONE sweep as a
                                             plain product and evaluates the whole-panel criterion; only if that fails does the
                                             fused fixed-point loop run (and the handle does not retry at or below that tol).
                                             0: always the loop.  Same fixed point, same criterion either way               */
#define SRI_ROD_OPT_FUSED_POSITION_RHS 101 /* 1 (default): in sri_rod_integrate the product that closes the quaternion solve also
                                              writes the position right-hand side R(q) e1 from its accumulators where the handle
                                              has that kernel variant (odd row counts in the 256-row tile geometry: N = 256, 512, 1024 ...); 0: a separate pass
                                              writes it.  Same bits                                                           */
#define SRI_ROD_OPT_FUSED_CRITERION 102 /* 1 (default): where that product can (operators of more than 64 rows in the paired k order — every
                                           integration operator the rod handle builds for N > 64), it also forms the whole-panel
                                           criterion 1/2 A(K) Q - G from its accumulators, and no separate pass reads Q and G back;
                                           0: the read-only pass.  The same fixed point; `last_update` is a rigorous bound either
                                           way (Cauchy-Schwarz over 64-row segments instead of whole columns: within ~30 % of each
                                           other, neither always the smaller)                                               */
int sri_rod_set_option(sri_rod_t *h, int option, long value);
/* The current value of a switch (apply options: as the integration operator's handle has them).  SRI_ROD_OPT_FUSED_CRITERION reads 2
 * when the criterion of the LAST solve on this handle came out of its closing product. */
int sri_rod_get_option(const sri_rod_t *h, int option, long *value);

/* Strain field K(X_i) = Phi(X_i) qe at the unknown nodes (utilities.h:49-67, main.cpp:69):
 * d_qe: 9 doubles per rod in main.cpp:187-195 order; dK: m x (3*rods), ldk >= m. */
int sri_rod_strain_field(sri_rod_t *h, const double *d_qe, size_t rods, double *dK, size_t ldk, void *hip_stream);

/* G_i = 1/2 A(K_i) Q_i per node (main.cpp:71-84). */
int sri_rod_quaternion_rhs(sri_rod_t *h, const double *dK, size_t ldk, const double *dQ, size_t ldq, size_t rods,
                           double *dG, size_t ldg, void *hip_stream);

/* One Picard iteration of the quaternion system in ONE launch (main.cpp:69-84 + 103-113, batched):
 *     Q      = Dn_NN^-1 G - (Dn_NN^-1 Dn_IN) q_base^T        (the integration apply with its initial-value term)
 *     G_next = 1/2 A(K) Q,   K = Phi(X_i) qe                  (formed in the apply's store stage, no K / Q panel)
 * dG, dG_next: m x (4*rods) panels (must not alias); d_q_base: 4*rods doubles, the base quaternion of every column;
 * d_tips (optional, 4*rods): tip-node values of Q, for convergence monitoring.  d_tips_prev + d_update (optional,
 * both or neither; need d_tips): the launch also accumulates max_col |tip - tips_prev| into *d_update as
 * max(*d_update, bits of that non-negative double) (NaN is stored as 0x7ff8000000000000); the caller zeroes it. */
int sri_rod_quaternion_step(sri_rod_t *h, const double *d_qe, const double *dG, size_t ldg, const double *d_q_base,
                            size_t rods, double *dG_next, size_t ldn, double *d_tips, const double *d_tips_prev,
                            uint64_t *d_update, void *hip_stream);

/* b_i = R(q_i) e1, updatePositionb (main.cpp:121-140). */
int sri_rod_position_rhs(sri_rod_t *h, const double *dQ, size_t ldq, size_t rods, double *dB, size_t ldb,
                         void *hip_stream);

/* integrateQuaternions (main.cpp:91-118) for a batch: solves (I_4 (x) Dn_NN - 1/2 A(K)) Q = -D_IN q_init by the
 * fixed-point form Q <- Dn_NN^-1 (1/2 A(K) Q - Dn_IN q_init^T) on the MFMA integration apply (one fused launch per
 * iteration, sri_rod_quaternion_step), until the largest update of any entry is <= tol (or max_iter:
 * SRI_ENOTCONVERGED).  q_init: 4 HOST doubles (NULL = 1,0,0,0).  dQ: m x (4*rods).  *last_update: a rigorous bound on
 * the next update of any entry of Q — with d = 1/2 A(K) Q - G over the whole panel, the smaller of max |d| (||Dn_NN^-1||_inf = 1)
 * and max_i ||row i of Dn_NN^-1||_2 * max_col ||d_col||_2 (Cauchy-Schwarz: the operator averages rounding noise).  *iters: fine
 * sweeps before the closing product (1 when the first sweep after a predicted start already met the criterion).  The loop is driven
 * on the device (tip-node monitor in the step's epilogue, a control kernel raising a stop word); the host waits on
 * the stream once per chunk of 8 iterations and once at the end, so the stream must not be capturing. */
int sri_rod_integrate_quaternions(sri_rod_t *h, const double *d_qe, size_t rods, const double *q_init, double *dQ,
                                  size_t ldq, double tol, uint32_t max_iter, uint32_t *iters, double *last_update,
                                  void *hip_stream);

/* integratePosition (main.cpp:145-176) for a batch: r = Dn_NN^-1 (R(q) e1 - Dn_IN r_init^T).  r_init: 3 HOST
 * doubles (NULL = 0,0,0).  dR: m x (3*rods). */
int sri_rod_integrate_positions(sri_rod_t *h, const double *dQ, size_t ldq, size_t rods, const double *r_init,
                                double *dR, size_t ldr, void *hip_stream);

/* Both, i.e. what main() computes for its one rod (main.cpp:197-201), for `rods` rods. */
int sri_rod_integrate(sri_rod_t *h, const double *d_qe, size_t rods, const double *q_init, const double *r_init,
                      double *dQ, size_t ldq, double *dR, size_t ldr, double tol, uint32_t max_iter,
                      uint32_t *iters, double *last_update, void *hip_stream);

/* ---------------------------------------------------------------------------------------- */
/* Multi-GPU: the one exchange step (the `sri_allgather` of §8b).  The rod batch */
/* shards by columns over the GPUs of a node — one process (or thread) per GPU, each with its   */
/* own sri_apply_t; the apply never communicates.  A communicator reassembles results with an   */
/* RCCL all-gather over xGMI, enqueued on the caller's HIP stream.  The reference has no        */
/* counterpart (main.cpp:181-205 is one thread, one rod); a C++ caller uses it as:              */
/*   rank 0: sri_comm_unique_id(id); ship the 128 bytes to the other ranks out of band          */
/*   all:    sri_comm_create(&c, id, rank, world, device);   (collective)                       */
/*           sri_apply_tips(...); sri_allgather_tips(c, d_tips, M_local, d_all, NULL, stream);   */
/* RCCL is loaded on first use (dlopen librccl.so.1); SRI_EUNSUPPORTED if it cannot be found.   */
/* ---------------------------------------------------------------------------------------- */
typedef struct sri_comm_handle sri_comm_t;
#define SRI_COMM_ID_BYTES 128

/* A fresh rendezvous token (ncclUniqueId), SRI_COMM_ID_BYTES bytes, produced by ONE rank. */
int sri_comm_unique_id(void *id);

/* Join the communicator `id` as `rank` of `world` on HIP device `device`; every rank must call. */
int sri_comm_create(sri_comm_t **out, const void *id, int rank, int world, int device);
int sri_comm_destroy(sri_comm_t *c);
int sri_comm_rank(const sri_comm_t *c, int *rank, int *world);

/* All-gather the tip-node value of every column (what sri_apply_tips / sri_integrate_tips emit): rank r's
 * `columns_local` doubles land in d_tips_all at the offset of its shard, rank order.  columns_per_rank: `world`
 * entries (ragged shards, rods % world != 0) or NULL when every rank holds columns_local. */
int sri_allgather_tips(sri_comm_t *c, const double *d_tips_local, size_t columns_local, double *d_tips_all,
                       const size_t *columns_per_rank, void *hip_stream);

/* All-gather whole output slabs: panels are column-major with leading dimension == rows, so every rank's slab is one
 * contiguous block and dY_all IS the global rows x (sum of columns) panel. */
int sri_allgather_panel(sri_comm_t *c, const double *dY_local, size_t rows, size_t columns_local, double *dY_all,
                        const size_t *columns_per_rank, void *hip_stream);

int sri_last_rccl_error(void);
const char *sri_rccl_strerror(int rccl_result);

/* ---------------------------------------------------------------------------------------- */
/* Introspection / diagnostics                                                              */
/* ---------------------------------------------------------------------------------------- */

/* Name of the kernel variant(s) the last sri_apply()/sri_integrate() on this handle launched (a string owned by the handle, e.g.
 * "sri_apply_mfma_f64<256x128,dma>", or "sri_apply_mfma_f64<256x128,dma,lean>|<64x64,dma>" when the column plan split the panel). */
const char *sri_apply_last_kernel(const sri_apply_t *h);

/* The same for the integration operator inside a rod handle: what its last launch was — a fused quaternion step
 * ("...,quat>"), or the plain integration that closes a solve ("none" for a solve that stayed in the one-launch on-chip
 * pipeline). */
const char *sri_rod_last_apply_kernel(const sri_rod_t *h);

/* The order in which a launch of this handle sums the products of one output element (see csrc/sri_internal.h):
 * *ascending = 1: k = 0, 1, 2, ... as main.cpp:172 does (operators of more than 512 rows other than Dn_NN^-1);
 * *ascending = 0: inside every block of 8 consecutive k, the even ones before the odd ones.  Either way every
 * multiply-add is fused (v_mfma_f64_16x16x4_f64 is a chain of four FMAs), and the result is a fixed function of the
 * inputs: no atomics, no run-to-run variation, the same bits for every batch size and device. */
int sri_apply_summation_order(const sri_apply_t *h, int *ascending);

const char *sri_strerror(int status);
int sri_last_hip_error(void);
int sri_abi_version(void);
int sri_device_count(int *count);

#ifdef __cplusplus
}
#endif
#endif /* SRI_CAPI_H */
