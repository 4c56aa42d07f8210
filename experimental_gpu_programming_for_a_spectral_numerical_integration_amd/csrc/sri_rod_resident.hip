// sri_rod_resident.hip — the whole rod pipeline of main.cpp:197-201 in ONE launch for small grids (N <= 64).
//
// For N - 1 = m <= 63 unknown nodes a rod's entire state fits on chip: the integration matrix Dn_NN^-1 (at most
// 32 KiB packed) lives in LDS for the workgroup, and every wave keeps the right-hand-side panel of ITS rods
// (4 columns per rod, K = 16*TM doubles per column) in its own LDS image.  A wave then runs the complete fixed-point
// iteration of the quaternion system (sri_rod.hip has the maths; same linear system as main.cpp:91-118)
//       G <- 1/2 A(K) (Dn_NN^-1 G - w q0^T)
// with v_mfma_f64_16x16x4_f64 reading X fragments from that image and writing the next right-hand side back into it,
// until max |G_new - G_old| <= tol — a bound on the next update of every entry of Q because ||Dn_NN^-1||_inf = 1
//  — then forms Q, the position right-hand side b = R(q) e1 (main.cpp:121-140) and
// r = Dn_NN^-1 b - w r0^T (main.cpp:145-176) the same way.  HBM sees 72 bytes of strain coefficients per rod coming in
// and the Q / r panels going out: nothing per iteration, no launch per iteration, no host round trip.
//
// Conventions shared with sri_apply.hip (see sri_internal.h): the X fragment is the MFMA's first operand, so a tile is
// accumulated transposed (lane (c, g) = (lane & 15, lane >> 4) holds row c of columns g, g+4, g+8, g+12); the column
// fed as MFMA column t is image column 4*(t&3) + (t>>2), so that those four registers are (w, x, y, z) of ONE rod at
// one node and 1/2 A(K) q is in-lane arithmetic; the 16-byte pieces of an image row are XOR-swizzled by
// (column >> 1) & 6 (conflict-free ds_read_b128 for that column order, tools/lds_bank_model.py); the initial-value term
// rides in the product: the host packs -w = -Dn_NN^-1 Dn_IN as operator column k = m and the image holds the base
// value of every column in slot k = m (TM = m / 16 + 1 row tiles, so that slot always exists: m = 16, 32, 48 — the grids
// of 2^p + 1 nodes — pay one nearly empty tile row for it).
// Waves of a workgroup share only the read-only operator: after one barrier they run independently.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#define SRI_DEVICE_HELPERS 1
#include "sri_internal.h"

namespace sri {

typedef double rd2 __attribute__((ext_vector_type(2)));
typedef double rd4 __attribute__((ext_vector_type(4)));

// A wave's LDS image is written element by element by every lane and read back as MFMA fragments that OTHER lanes wrote.  The
// hardware keeps one wave's LDS operations in order, but the compiler reasons per lane: a lane's own writes and its fragment
// reads are provably disjoint addresses, so without this it may move the reads above the writes (seen with the coarse start at
// TN > 1: the coarse loop iterated on a stale image).  No instruction is emitted.
__device__ __forceinline__ void wave_image_fence() { asm volatile("" ::: "memory"); }

#define RES_HIP(call)                                            \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) {                                     \
      sri::record_hip_error(static_cast<int>(e__));              \
      (void)hipGetLastError();                                   \
      return SRI_EHIP;                                           \
    }                                                            \
  } while (0)

struct ResidentArgs {
  const double* apack;     // [k chunk][row tile][k pair][lane][2], -w folded in as column k = m
  const double* p1;        // P1 = 2X - 1 at the unknown nodes, zero padded to 16*TM
  const double* qe;        // 9 per rod
  double* Q;               // m x (4*rods)
  double* R;               // m x (3*rods) or null (quaternions only)
  double* stats;           // 2 per workgroup: {most iterations of any wave, largest final update (NaN if any)}
  size_t rods, ldq, ldr;
  double q0[4], r0[3];
  double tol;
  uint32_t m, max_iter;
  // coarse start (grids of more than 16 nodes): the packed integration operator of the coarse grid (16 TMC nodes: TMC^2 tiles
  // of 256 doubles, -w folded in as column k = 16 TMC - 1), the interpolation matrix from its nodes to the m unknown fine
  // nodes as TM x TMC tiles of the same fragment layout, and P1 at the coarse nodes; null = plain start Q^0 = q0
  const double* cpack;
  const double* p1c;
};

// TM row tiles (m <= 16*TM - 1), TN column tiles of 16 columns = 4 rods per wave task, NW waves per workgroup.
template <int TM, int TN, int NW>
struct ResGeo {
  static constexpr int A_BYTES = TM * TM * 2048;
  static constexpr int CHUNK_BYTES = TN * 16 * 128;            // one k chunk of a wave's image
  static constexpr int IMG_BYTES = TM * CHUNK_BYTES;
  // coarse start: a 16-node grid under fine grids of up to 32 nodes, a 32-node grid (TMC = 2) under larger ones
  static constexpr int TMC = TM >= 3 ? 2 : 1;
  static constexpr int COP_BYTES = TMC * TMC * 2048;           // coarse operator tiles, [kc][tmc][j] like the fine one
  static constexpr int C_BYTES = TM >= 2 ? COP_BYTES + TM * TMC * 2048 : 0;   // ... + interpolation tiles [tm][kc][j]
  static constexpr int C_OFFSET = A_BYTES + NW * IMG_BYTES + NW * 16;
  static constexpr int LDS_BYTES = C_OFFSET + C_BYTES;
  static constexpr int THREADS = 64 * NW;
  static constexpr int RODS_PER_TASK = 4 * TN;
  static_assert(LDS_BYTES <= 160 * 1024, "operator + wave images exceed a CU");
};

// (second launch bound = waves per SIMD the register budget must allow: four for the 34-KiB geometry, which fits four
// workgroups per CU; two elsewhere)
template <int TM, int TN, int NW>
__global__ __launch_bounds__(64 * NW, TM == 1 ? 4 : 2) void rod_resident_kernel(const ResidentArgs p) {
  using G = ResGeo<TM, TN, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;

  // operator -> LDS, once per workgroup (the packed layout is the read layout: a linear copy)
  for (int i = tid; i < G::A_BYTES / 16; i += G::THREADS)
    reinterpret_cast<rd2*>(smem)[i] = reinterpret_cast<const rd2*>(p.apack)[i];
  // Coarse start (nested iteration, as the streaming path's predictor — sri_rod.hip — but on chip): plain Picard from
  // Q = q0 needs ~22 sweeps because its error decays like (|K|/2)^k / k!; the strain fields are quadratic polynomials
  // (utilities.h:49-67), the solution is analytic, and a 16-node grid resolves it to ~1e-9, a 32-node grid to rounding.  So
  // the wave first runs the SAME fixed point on a coarse grid in the first chunks of its image — 16 nodes (one operator
  // tile) under fine grids of up to 32 nodes, 32 nodes (2 x 2 tiles: a quarter of a 64-node sweep, and the fine loop then
  // starts at rounding level) under larger ones — interpolates Q to the fine nodes with one more MFMA product (the
  // interpolation matrix packed like an operator column block) and starts the fine loop from G^0 = 1/2 A(K) P [Q_c; q0].  The fine loop, its stopping criterion and
  // everything after it are unchanged: the result is still the fixed point of the FINE system.
  const bool coarse = TM >= 2 && p.cpack != nullptr;          // uniform over the grid
  char* const cop = smem + G::C_OFFSET;                       // coarse operator (2 KiB), then TM interpolation tiles
  if constexpr (TM >= 2) {
    if (coarse)
      for (int i = tid; i < G::C_BYTES / 16; i += G::THREADS)
        reinterpret_cast<rd2*>(cop)[i] = reinterpret_cast<const rd2*>(p.cpack)[i];
  }
  __syncthreads();

  char* img = smem + G::A_BYTES + wid * G::IMG_BYTES;
  const int cs = 4 * (c & 3) + (c >> 2);                       // image column fed as MFMA column c
  // byte offset, inside a k chunk of the image, of the fragment this lane reads for column tile tn, k pair j
  auto frag_off = [&](int tn, int j) __attribute__((always_inline)) {
    const int col = tn * 16 + cs;
    return col * 128 + (((4 * j + g) ^ ((col >> 1) & 6)) << 4);
  };
  // ... and of element (k = c of chunk tm) of column `col`, which this lane writes
  auto elem_off = [&](int col) __attribute__((always_inline)) {
    return col * 128 + ((((c >> 1)) ^ ((col >> 1) & 6)) << 4) + (c & 1) * 8;
  };

  double p1[TM], p2[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    p1[tm] = p.p1[tm * 16 + c];
    p2[tm] = ((2 * 1 + 1) * p1[tm] * p1[tm] - 1 * 1.0) / (1 + 1);            // as strain_kernel (sri_rod.hip)
  }

  constexpr int TMC = G::TMC;
  double p1c[TMC], p2c[TMC];                                   // P1, P2 at coarse node 16 tmc + c
#pragma unroll
  for (int tmc = 0; tmc < TMC; ++tmc) {
    p1c[tmc] = 0.0;
    p2c[tmc] = 0.0;
    if constexpr (TM >= 2) {
      if (coarse) {
        p1c[tmc] = p.p1c[tmc * 16 + c];
        p2c[tmc] = ((2 * 1 + 1) * p1c[tmc] * p1c[tmc] - 1 * 1.0) / (1 + 1);
      }
    }
  }
  const size_t ntasks = (p.rods + G::RODS_PER_TASK - 1) / G::RODS_PER_TASK;
  uint32_t worst_it = 0;
  double worst_upd = 0.0;

  // Y = A * image for this wave's columns: acc[tm][tn][r] = Y[16 tm + c][16 tn + 4 g + r]
  rd4 acc[TM][TN];
  auto product = [&]() __attribute__((always_inline)) {
    wave_image_fence();
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = rd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc < TM; ++kc)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        rd2 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          a[tm] = *reinterpret_cast<const rd2*>(smem + ((kc * TM + tm) * 2 + j) * 1024 + lane * 16);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const rd2*>(img + kc * G::CHUNK_BYTES + frag_off(tn, j));
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[tn][e], a[tm][e], acc[tm][tn], 0, 0, 0);
      }
  };
  // f64 compares and selects run at the rate of the MFMA pipe they share, so the convergence monitor uses one v_max_f64
  // per value (which DROPS NaNs) and NaNs are caught once per task instead, on the final iterate.
  auto wave_max = [&](double v) __attribute__((always_inline)) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = __builtin_fmax(v, __shfl_xor(v, off));
    return v;
  };

  for (size_t task = static_cast<size_t>(blockIdx.x) * NW + wid; task < ntasks; task += static_cast<size_t>(gridDim.x) * NW) {
    const size_t rod0 = task * G::RODS_PER_TASK + g;              // this lane's rod in column tile tn is rod0 + 4 tn
    // K/2 at this lane's (row, rod) pairs — invariant over the iteration
    double hk[TM][TN][3];
    double hkc[TMC][TN][3];                                      // ... and at the coarse nodes (coarse start only)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      size_t rod = rod0 + 4 * tn;
      rod = rod < p.rods ? rod : p.rods - 1;                      // tail task: recompute a valid rod, never store it
      const double* q = p.qe + rod * 9;
      double q9[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) q9[i] = q[i];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          double s = q9[a * 3 + 0];
          s = s + p1[tm] * q9[a * 3 + 1];
          s = s + p2[tm] * q9[a * 3 + 2];
          hk[tm][tn][a] = 0.5 * s;
        }
#pragma unroll
      for (int tmc = 0; tmc < TMC; ++tmc)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          double s = q9[a * 3 + 0];
          s = s + p1c[tmc] * q9[a * 3 + 1];
          s = s + p2c[tmc] * q9[a * 3 + 2];
          hkc[tmc][tn][a] = 0.5 * s;
        }
    }
    bool predicted = false;
    if constexpr (TM >= 2) {
      if (coarse) {
        // ---- the same fixed point on the coarse grid (mc = 16 TMC - 1 unknown nodes), in chunks 0..TMC-1 of the image:
        //      rows 0..mc-1 = G_c, row mc = q0 ------------------------------------------------------------------------
        constexpr int mc = 16 * TMC - 1;
        rd4 accc[TMC][TN];
        auto product_c = [&]() __attribute__((always_inline)) {
          wave_image_fence();
#pragma unroll
          for (int tmc = 0; tmc < TMC; ++tmc)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) accc[tmc][tn] = rd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kc = 0; kc < TMC; ++kc)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              rd2 a[TMC], b[TN];
#pragma unroll
              for (int tmc = 0; tmc < TMC; ++tmc)
                a[tmc] = *reinterpret_cast<const rd2*>(cop + ((kc * TMC + tmc) * 2 + j) * 1024 + lane * 16);
#pragma unroll
              for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const rd2*>(img + kc * G::CHUNK_BYTES + frag_off(tn, j));
#pragma unroll
              for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int tmc = 0; tmc < TMC; ++tmc)
#pragma unroll
                  for (int tn = 0; tn < TN; ++tn)
                    accc[tmc][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[tn][e], a[tmc][e], accc[tmc][tn], 0, 0, 0);
            }
        };
#pragma unroll
        for (int tmc = 0; tmc < TMC; ++tmc) {
          const int row = tmc * 16 + c;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            double gq[4];
            quat_rhs_half(hkc[tmc][tn][0], hkc[tmc][tn][1], hkc[tmc][tn][2], p.q0[0], p.q0[1], p.q0[2], p.q0[3], gq);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *reinterpret_cast<double*>(img + tmc * G::CHUNK_BYTES + elem_off(tn * 16 + 4 * g + r)) = row < mc ? gq[r] : p.q0[r];
          }
        }
        const uint32_t coarse_budget = p.max_iter < 100u ? p.max_iter : 100u;
        for (uint32_t itc = 0; itc < coarse_budget; ++itc) {
          product_c();
          double local = 0.0;
#pragma unroll
          for (int tmc = 0; tmc < TMC; ++tmc) {
            if (tmc * 16 + c < mc) {
#pragma unroll
              for (int tn = 0; tn < TN; ++tn) {
                double gq[4];
                quat_rhs_half(hkc[tmc][tn][0], hkc[tmc][tn][1], hkc[tmc][tn][2], accc[tmc][tn][0], accc[tmc][tn][1], accc[tmc][tn][2],
                              accc[tmc][tn][3], gq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  double* slot = reinterpret_cast<double*>(img + tmc * G::CHUNK_BYTES + elem_off(tn * 16 + 4 * g + r));
                  local = __builtin_fmax(local, fabs(gq[r] - *slot));
                  *slot = gq[r];
                }
              }
            }
          }
          if (wave_max(local) <= 0.25 * p.tol) break;           // wave-uniform
        }
        product_c();                                             // Q_c at the unknown coarse nodes; row mc of the image still holds q0
#pragma unroll
        for (int tmc = 0; tmc < TMC; ++tmc) {
          if (tmc * 16 + c < mc) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *reinterpret_cast<double*>(img + tmc * G::CHUNK_BYTES + elem_off(tn * 16 + 4 * g + r)) = accc[tmc][tn][r];
          }
        }
        // ---- Q at the fine nodes: acc[tm][tn] = P_tile[tm][.] . [Q_c; q0]  (TMC k chunks of the "operator" P) ------------
        wave_image_fence();
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = rd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kc = 0; kc < TMC; ++kc)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            rd2 a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
              a[tm] = *reinterpret_cast<const rd2*>(cop + G::COP_BYTES + ((tm * TMC + kc) * 2 + j) * 1024 + lane * 16);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const rd2*>(img + kc * G::CHUNK_BYTES + frag_off(tn, j));
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                  acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[tn][e], a[tm][e], acc[tm][tn], 0, 0, 0);
          }
        wave_image_fence();                                      // the coarse chunks have been read; the fine image is written next
        predicted = true;
      }
    }
    // image <- G^0 = 1/2 A(K) Q^0 at the unknown nodes (Q^0 = q0, or the interpolated coarse solution in acc); slot k = m <- q0
    // (the base value of every column); rest 0
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const uint32_t row = tm * 16 + c;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        double gq[4] = {0.0, 0.0, 0.0, 0.0};
        if (row < p.m) {
          if (predicted) quat_rhs_half(hk[tm][tn][0], hk[tm][tn][1], hk[tm][tn][2], acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3], gq);
          else quat_rhs_half(hk[tm][tn][0], hk[tm][tn][1], hk[tm][tn][2], p.q0[0], p.q0[1], p.q0[2], p.q0[3], gq);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = row < p.m ? gq[r] : (row == p.m ? p.q0[r] : 0.0);
          *reinterpret_cast<double*>(img + tm * G::CHUNK_BYTES + elem_off(tn * 16 + 4 * g + r)) = v;
        }
      }
    }

    // ---- fixed point ------------------------------------------------------------------------------
    uint32_t it = 0;
    double upd = INFINITY;
    while (it < p.max_iter) {
      product();                                                  // acc = Q^{it+1} (initial-value term included)
      double local = 0.0;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const uint32_t row = tm * 16 + c;
        if (row < p.m) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            double gq[4];
            quat_rhs_half(hk[tm][tn][0], hk[tm][tn][1], hk[tm][tn][2], acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2],
                          acc[tm][tn][3], gq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double* slot = reinterpret_cast<double*>(img + tm * G::CHUNK_BYTES + elem_off(tn * 16 + 4 * g + r));
              local = __builtin_fmax(local, fabs(gq[r] - *slot));
              *slot = gq[r];
            }
          }
        }
      }
      ++it;
      upd = wave_max(local);
      if (upd <= p.tol) break;                                    // wave-uniform
    }

    // ---- Q = Dn_NN^-1 G - w q0^T, stored; then b = R(q) e1 into the image (4th column of a rod: 0) -------------
    product();
    {
      bool bad = false;                                           // a NaN anywhere in the task's state poisons Q
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const double t = (acc[tm][tn][0] + acc[tm][tn][1]) + (acc[tm][tn][2] + acc[tm][tn][3]);
          bad = bad || (t != t);
        }
      if (__ballot(bad) != 0ull) upd = __builtin_nan("");
    }
    worst_it = it > worst_it ? it : worst_it;
    worst_upd = (upd > worst_upd || upd != upd) ? upd : worst_upd;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const uint32_t row = tm * 16 + c;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const size_t rod = rod0 + 4 * tn;
        const double w = acc[tm][tn][0], x = acc[tm][tn][1], y = acc[tm][tn][2], z = acc[tm][tn][3];
        if (row < p.m && rod < p.rods) {
          double* qc = p.Q + rod * 4 * p.ldq + row;
          qc[0] = w; qc[p.ldq] = x; qc[2 * p.ldq] = y; qc[3 * p.ldq] = z;
        }
        if (p.R != nullptr) {
          // main.cpp:136 q.toRotationMatrix() * e1 (Eigen: no normalisation), as position_rhs_kernel (sri_rod.hip)
          const double ty = 2 * y, tz = 2 * z;
          const double twy = ty * w, twz = tz * w, txy = ty * x, txz = tz * x, tyy = ty * y, tzz = tz * z;
          const double b[4] = {1 - (tyy + tzz), txy + twz, txz - twy, 0.0};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double v = row < p.m ? b[r] : (row == p.m && r < 3 ? p.r0[r] : 0.0);
            *reinterpret_cast<double*>(img + tm * G::CHUNK_BYTES + elem_off(tn * 16 + 4 * g + r)) = v;
          }
        }
      }
    }
    if (p.R != nullptr) {
      product();                                                  // r = Dn_NN^-1 b - w r0^T  (main.cpp:172)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const uint32_t row = tm * 16 + c;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const size_t rod = rod0 + 4 * tn;
          if (row < p.m && rod < p.rods) {
            double* rc = p.R + rod * 3 * p.ldr + row;
            rc[0] = acc[tm][tn][0]; rc[p.ldr] = acc[tm][tn][1]; rc[2 * p.ldr] = acc[tm][tn][2];
          }
        }
      }
    }
  }

  // per-workgroup statistics (every wave arrives here: the task loop has no barrier inside)
  double* wstat = reinterpret_cast<double*>(smem + G::A_BYTES + NW * G::IMG_BYTES);
  if (lane == 0) {
    wstat[2 * wid] = static_cast<double>(worst_it);
    wstat[2 * wid + 1] = worst_upd;
  }
  __syncthreads();
  if (tid == 0) {
    double its = 0.0, u = 0.0;
    for (int w = 0; w < NW; ++w) {
      its = wstat[2 * w] > its ? wstat[2 * w] : its;
      const double v = wstat[2 * w + 1];
      u = (v > u || v != v) ? v : u;
    }
    p.stats[2 * blockIdx.x] = its;
    p.stats[2 * blockIdx.x + 1] = u;
  }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
struct ResidentRod {
  int device = 0;
  uint32_t n = 0, m = 0;
  int tm = 0;
  int num_cus = 0;
  double* d_apack = nullptr;
  double* d_p1 = nullptr;
  double* d_cpack = nullptr;      // coarse start (grids of more than 16 nodes): coarse operator + interpolation tiles
  double* d_p1c = nullptr;
  double* d_stats = nullptr;
  double* h_stats = nullptr;      // pinned
  uint32_t pending_blocks = 0;    // resident_integrate(..., defer_stats): entries of h_stats the caller collects after ITS wait
  uint32_t cap_blocks = 0;
};

bool resident_supported(uint32_t m) { return m >= 2 && m <= 63; }

template <int TM, int TN, int NW>
static int resident_launch(ResidentRod* h, ResidentArgs args, hipStream_t stream, uint32_t* blocks_out) {
  using G = ResGeo<TM, TN, NW>;
  const auto kern = &rod_resident_kernel<TM, TN, NW>;
  // blocks per CU of this geometry, per device: written once under the lock (handles of several devices may take their
  // first launch from different host threads), read lock-free afterwards
  static std::atomic<int> per_cu[16] = {};
  static std::mutex prepare_lock;
  if (h->device < 0 || h->device >= 16) return SRI_EUNSUPPORTED;
  int resident_blocks = per_cu[h->device].load(std::memory_order_acquire);
  if (resident_blocks == 0) {
    std::lock_guard<std::mutex> guard(prepare_lock);
    resident_blocks = per_cu[h->device].load(std::memory_order_acquire);
    if (resident_blocks == 0) {
      RES_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
      int nb = 0;
      RES_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, G::THREADS, G::LDS_BYTES));
      resident_blocks = nb < 1 ? 1 : nb;
      per_cu[h->device].store(resident_blocks, std::memory_order_release);
    }
  }
  const size_t ntasks = (args.rods + G::RODS_PER_TASK - 1) / G::RODS_PER_TASK;
  const size_t wgs_needed = (ntasks + NW - 1) / NW;
  size_t grid = static_cast<size_t>(h->num_cus) * resident_blocks;
  grid = wgs_needed < grid ? wgs_needed : grid;
  if (grid > h->cap_blocks) return SRI_EUNSUPPORTED;
  *blocks_out = static_cast<uint32_t>(grid);
  RES_HIP(sri::launch_kernel((rod_resident_kernel<TM, TN, NW>), dim3(static_cast<unsigned>(grid)), dim3(G::THREADS), G::LDS_BYTES, stream, args));
  return SRI_OK;
}

int resident_destroy(ResidentRod* h) {
  if (h == nullptr) return SRI_OK;
  if (h->d_apack) (void)hipFree(h->d_apack);
  if (h->d_p1) (void)hipFree(h->d_p1);
  if (h->d_cpack) (void)hipFree(h->d_cpack);
  if (h->d_p1c) (void)hipFree(h->d_p1c);
  if (h->d_stats) (void)hipFree(h->d_stats);
  if (h->h_stats) (void)hipHostFree(h->h_stats);
  delete h;
  return SRI_OK;
}

static int resident_create_parts(ResidentRod* h, uint32_t n, int device, int num_cus, const double* A_fine, const double* w_fine, uint32_t n_fine);

// Caller has made `device` current.
int resident_create(uint32_t n, int device, int num_cus, ResidentRod** out, const double* A_fine, const double* w_fine, uint32_t n_fine) {
  *out = nullptr;
  if (!resident_supported(n - 1)) return SRI_EUNSUPPORTED;
  ResidentRod* h = new (std::nothrow) ResidentRod();
  if (h == nullptr) return SRI_ENOMEM;
  int rc;
  try {                                                         // (host vectors: no exception may reach the C ABI above)
    rc = resident_create_parts(h, n, device, num_cus, A_fine, w_fine, n_fine);
  } catch (const std::bad_alloc&) {
    rc = SRI_ENOMEM;
  }
  if (rc != SRI_OK) {
    resident_destroy(h);
    return rc;
  }
  *out = h;
  return SRI_OK;
}

// on failure the caller destroys `h`
static int resident_create_parts(ResidentRod* h, uint32_t n, int device, int num_cus, const double* A_fine, const double* w_fine, uint32_t n_fine) {
  const uint32_t m = n - 1;
  h->device = device;
  h->n = n;
  h->m = m;
  h->tm = static_cast<int>(m / 16 + 1);
  h->num_cus = num_cus;
  const int TM = h->tm;
  std::vector<double> A(static_cast<size_t>(m) * m), Din(m), w(m, 0.0), x(n);
  int rc = cheb_integration(n, A.data(), m, Din.data());
  if (rc != SRI_OK) return rc;
  for (uint32_t k = 0; k < m; ++k)
    for (uint32_t i = 0; i < m; ++i) w[i] = w[i] + A[i + static_cast<size_t>(k) * m] * Din[k];   // as sri_apply_create
  if (A_fine != nullptr && w_fine != nullptr && n_fine > n) {    // this handle starts the fixed point of an n_fine-node grid
    rc = coarse_view_of_fine_operator(A_fine, n_fine - 1, w_fine, n_fine, n, A, w);
    if (rc != SRI_OK) return rc;
  }
  cheb_points(n, 1.0, x.data());
  std::vector<double> packed(static_cast<size_t>(TM) * TM * 256, 0.0), p1(static_cast<size_t>(TM) * 16, 0.0);
  size_t o = 0;
  for (int kc = 0; kc < TM; ++kc)
    for (int tm = 0; tm < TM; ++tm)
      for (int j = 0; j < 2; ++j)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 2; ++e, ++o) {
            const uint32_t row = tm * 16 + (l & 15);
            const uint32_t k = kc * 16 + k_of_paired(j, l >> 4, e);
            if (row < m && k < m) packed[o] = A[row + static_cast<size_t>(k) * m];
            else if (row < m && k == m) packed[o] = -w[row];
          }
  for (uint32_t i = 0; i < m; ++i) p1[i] = (2 * x[i] - (1.0 + 0.0)) / (1.0 - 0.0);
  h->cap_blocks = static_cast<uint32_t>(num_cus) * 16u;
  auto fail = [&](int code) { return code; };
  if (hipMalloc(&h->d_apack, packed.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
  if (hipMalloc(&h->d_p1, p1.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
  if (hipMalloc(&h->d_stats, 2 * sizeof(double) * h->cap_blocks) != hipSuccess) return fail(SRI_ENOMEM);
  if (hipHostMalloc(reinterpret_cast<void**>(&h->h_stats), 2 * sizeof(double) * h->cap_blocks, hipHostMallocDefault) != hipSuccess)
    return fail(SRI_ENOMEM);
  if (hipMemcpy(h->d_apack, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
  if (hipMemcpy(h->d_p1, p1.data(), p1.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
  if (TM >= 2) {
    // coarse start: the fine operator as a coarse grid sees it (16 nodes under fine grids of up to 32, else 32: TMC tiles a
    // side, -w_c folded in as column k = mc), and the interpolation matrix from the coarse nodes (mc unknown + the base node,
    // in that order: both grids descend from the tip to the base, chebyshev_differentiation.h:26) to the m unknown fine
    // nodes, as TM x TMC tiles
    const int TMC = TM >= 3 ? 2 : 1;
    const uint32_t nc = 16u * TMC, mc = nc - 1;
    std::vector<double> Ac, wc, xc(nc), Pm(static_cast<size_t>(n) * nc);
    rc = coarse_view_of_fine_operator(A.data(), m, w.data(), n, nc, Ac, wc);   // what the fine loop will apply, at the coarse nodes
    if (rc == SRI_OK) rc = cheb_interp(nc, n, Pm.data(), n);
    if (rc != SRI_OK) return fail(rc);
    cheb_points(nc, 1.0, xc.data());
    std::vector<double> cpack(static_cast<size_t>(TMC * TMC + TM * TMC) * 256, 0.0), p1c(nc, 0.0);
    o = 0;
    for (int kc = 0; kc < TMC; ++kc)
      for (int tmc = 0; tmc < TMC; ++tmc)
        for (int j = 0; j < 2; ++j)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 2; ++e, ++o) {
              const uint32_t row = tmc * 16 + (l & 15), k = kc * 16 + k_of_paired(j, l >> 4, e);
              if (row < mc && k < mc) cpack[o] = Ac[row + static_cast<size_t>(k) * mc];
              else if (row < mc && k == mc) cpack[o] = -wc[row];
            }
    for (int tm = 0; tm < TM; ++tm)
      for (int kc = 0; kc < TMC; ++kc)
        for (int j = 0; j < 2; ++j)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 2; ++e, ++o) {
              const uint32_t row = tm * 16 + (l & 15), k = kc * 16 + k_of_paired(j, l >> 4, e);
              if (row < m) cpack[o] = Pm[row + static_cast<size_t>(k) * n];
            }
    for (uint32_t i = 0; i < nc; ++i) p1c[i] = (2 * xc[i] - (1.0 + 0.0)) / (1.0 - 0.0);
    if (hipMalloc(&h->d_cpack, cpack.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
    if (hipMalloc(&h->d_p1c, p1c.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
    if (hipMemcpy(h->d_cpack, cpack.data(), cpack.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
    if (hipMemcpy(h->d_p1c, p1c.data(), p1c.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
  }
  return SRI_OK;
}

static void fold_stats(const ResidentRod* h, uint32_t blocks, uint32_t* iters_out, double* last_update_out) {
  double its = 0.0, upd = 0.0;
  for (uint32_t b = 0; b < blocks; ++b) {
    its = h->h_stats[2 * b] > its ? h->h_stats[2 * b] : its;
    const double v = h->h_stats[2 * b + 1];
    upd = (v > upd || v != v) ? v : upd;
  }
  if (iters_out) *iters_out = static_cast<uint32_t>(its);
  if (last_update_out) *last_update_out = upd;
}

// The statistics of a launch enqueued with defer_stats (iterations, last update: NaN if any rod produced one): copied and waited for
// HERE — the rod predictor only asks when the bound it reads at the end of its sequence is NaN
int resident_collect(ResidentRod* h, uint32_t* iters_out, double* last_update_out, void* stream_) {
  if (h == nullptr) return SRI_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  RES_HIP(hipMemcpyAsync(h->h_stats, h->d_stats, 2 * sizeof(double) * h->pending_blocks, hipMemcpyDeviceToHost, stream));
  RES_HIP(hipStreamSynchronize(stream));
  fold_stats(h, h->pending_blocks, iters_out, last_update_out);
  h->pending_blocks = 0;
  return SRI_OK;
}

// Quaternions (dR == nullptr) or quaternions + positions of `rods` rods; synchronises the stream once (statistics) — unless
// defer_stats: then the copy of the statistics is only enqueued (the rod predictor: no host round trip between the coarse solve
// and the prolongation; resident_collect after the solve's one wait).
int resident_integrate(ResidentRod* h, const double* d_qe, size_t rods, const double* q_init, const double* r_init, double* dQ,
                       size_t ldq, double* dR, size_t ldr, double tol, uint32_t max_iter, uint32_t* iters_out,
                       double* last_update_out, void* stream_, bool coarse_start, bool defer_stats) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ResidentArgs args;
  args.apack = h->d_apack;
  args.p1 = h->d_p1;
  args.qe = d_qe;
  args.Q = dQ;
  args.R = dR;
  args.stats = h->d_stats;
  args.rods = rods;
  args.ldq = ldq;
  args.ldr = ldr;
  for (int i = 0; i < 4; ++i) args.q0[i] = q_init[i];
  for (int i = 0; i < 3; ++i) args.r0[i] = r_init ? r_init[i] : 0.0;
  args.tol = tol;
  args.m = h->m;
  args.max_iter = max_iter;
  args.cpack = coarse_start ? h->d_cpack : nullptr;            // (null for grids of 16 nodes or fewer: nothing coarser to start from)
  args.p1c = h->d_p1c;
  uint32_t blocks = 0;
  int rc;
  switch (h->tm) {
    case 1: rc = resident_launch<1, 4, 4>(h, args, stream, &blocks); break;    // 16 rods per wave, 34 KiB: several WG per CU
    case 2: rc = resident_launch<2, 4, 4>(h, args, stream, &blocks); break;    // 72 KiB: 2 WG per CU
    case 3: rc = resident_launch<3, 2, 8>(h, args, stream, &blocks); break;    // 114 KiB: 1 WG (8 waves) per CU
    default: rc = resident_launch<4, 1, 8>(h, args, stream, &blocks); break;   //  96 KiB: 1 WG (8 waves) per CU
  }
  if (rc != SRI_OK) return rc;
  if (defer_stats) {
    h->pending_blocks = blocks;
    return SRI_OK;
  }
  RES_HIP(hipMemcpyAsync(h->h_stats, h->d_stats, 2 * sizeof(double) * blocks, hipMemcpyDeviceToHost, stream));
  RES_HIP(hipStreamSynchronize(stream));
  double upd = 0.0;
  fold_stats(h, blocks, iters_out, &upd);
  if (last_update_out) *last_update_out = upd;
  return (upd <= tol) ? SRI_OK : SRI_ENOTCONVERGED;
}

}  // namespace sri
