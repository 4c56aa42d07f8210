// sri_column_plan.h — which tile geometry takes which columns of a panel (pure C++, no HIP: tests/cpp/column_plan_check.cpp)
//
// Every launch of sri::apply_kernel is persistent: as many workgroups as the chip keeps resident, each walking its share of
// the tiles.  A launch therefore costs WHOLE rounds of tiles — 1.1 rounds of 256 x 128 tiles take as long as 2 — and a panel
// of a few thousand columns leaves most CUs idle for a whole tile time (measured, N = 256: 4 096 columns take 61.7 us in
// 256 x 128 tiles, 34.0 us in 128 x 128 tiles and 13.2 us in 64 x 64 tiles; 49 152 columns 128.5 / 124.7 / 101.4 us:
// profiles/r04_geometry_by_columns.log).  The smaller geometries cost 10-15 % of efficiency on full rounds but their rounds
// are short.  So a launch is planned per call: the operator's own ("primary") geometry takes the columns of its full rounds,
// one of the finer geometries of the same operator ("alternates": every output element is the same chain of fused
// multiply-adds in the same k order, so the results are the same bits) takes the rest — or everything, for a small panel.
//
// The plan comes from a cost model, not from a search on the device (launch_us() below): a launch of at most one round takes
// t(workgroups sharing a CU) per 16-k chunk; several rounds of the large geometries take whole rounds, the last one as full as
// its tiles make it; the 64 x 64 geometry, five workgroups to a CU, works through a CU's share of the tiles like a queue; plus a
// launch overhead.  The table was measured at N = 128 ... 384 on an MI355X (round 4).  The model only has to rank candidates
// whose times differ by more than its own error: a candidate replaces "everything in the primary geometry" only if the model
// sees at least 5 % and 3 us to win (measured against the plan switched off: profiles/r04_column_plan_probe.log).
#ifndef SRI_COLUMN_PLAN_H
#define SRI_COLUMN_PLAN_H

#include <cmath>
#include <cstddef>
#include <cstdint>

namespace sri {
namespace plan {

// ids as sri::GeoId (sri_apply_kernel.h)
enum : int { FLAT = 0, MID = 1, TALL = 2, WIDE = 3, WIDE32 = 4, TALL_STRIP = 5, G96 = 6, G192 = 7, G160 = 8, G224 = 9, TALL_MINI = 10, MID_MINI = 11, FLAT_MINI = 12, GEO_COUNT = 13 };

// ---- rows: the operator's own geometry, and the parts it is cut into (sri_apply.hip's create_from_matrix) ----------------------
// Rows are tiled in blocks of 64 / 128 / 256 and a partly empty block costs a full one.  Above 256 rows the 128-row geometry is
// the better choice whenever it saves a half-empty 256-row block (round 3).
inline int own_geometry(uint32_t rows_out) {
  const bool half_empty_last_block = rows_out > 256 && (rows_out - 1) % 256 < 128;
  return rows_out <= 64 ? FLAT : ((rows_out <= 128 || half_empty_last_block) ? MID : TALL);
}

struct RowPart {
  int geo = TALL;
  uint32_t row0 = 0, rows = 0, row_blocks = 0;
};

// Parts (round 4), by the rows behind the last full 256-row block: 65-96 rows are ONE part in the 96 x 128 geometry, 129-160 rows one in the 160 x 64 geometry, 161-192 rows
// one in the 192 x 64 geometry, 193-224 rows (alone) one in the 224 x 64 geometry — a row block of the operator's own height (plain and rank-1-epilogue launches; the fused step and
// sri_apply_many stay on the operator's own 128- / 256-row geometry).  With U = ceil(rows / 64) units, U = 4a + 1 (257-320, 513-576, 1025-1088, ... rows) is 256 a rows in the
// 256-row geometry plus a last part of 1-64 rows — as a 16-row strip of the last 256-row block if it is at most 16 rows
// (then the first a - 1 blocks are a part of their own), else in the 32 x 256 / 64 x 256 geometry.  Returns the number of parts
// (0: the operator stays whole).
inline int row_parts(uint32_t rows, bool allow_strip, RowPart out[2], bool extended = true, bool allow_mini = true) {
  int np = 0;
  // the rows behind the last full 256-row block: 1 ... 256 of them (all of the operator's rows when it has at most 256)
  const uint32_t full = rows > 256 ? (rows - 1) / 256 * 256 : 0, rest = rows - full;
  int last = -1;                                                   // geometry of a last part of `rest` rows (-1: none)
  if (rest > 64 && rest <= 68 && allow_mini && allow_strip) last = FLAT_MINI;        // a 64-row block + its 1-4 leftover rows on the 4x4x4 MFMA (N = 65) ...
  else if (rest > 64 && rest <= 96) last = G96;                    // ... one row block of 96 ...
  else if (rest > 128 && rest <= 132 && allow_mini && allow_strip) last = MID_MINI;   // ... a 128-row block + its 1-4 leftover rows on the 4x4x4 MFMA (N = 129) ...
  else if (rest > 128 && rest <= 160) last = G160;                 // ... of 160 (160 x 64 tiles) ...
  else if (rest > 160 && rest <= 192) last = G192;                 // ... or of 192 (192 x 64 tiles)
  if (full == 0) {                                                 // at most 256 rows: one block of the operator's own height, in one pass
    if (rest > 192 && rest <= 224) last = G224;                    // (behind full blocks a 256-row block that is 75-87 % full does as well)
    if (last >= 0) out[np++] = RowPart{last, 0, rows, 1};
    return np;
  }
  if (rest <= 16 && allow_strip) {
    // (at most four rows — ONE on every N = 2^p + 1 grid — ride on the 4x4x4 MFMA: the mini strip, a quarter of the strip's cost)
    if (full > 256) out[np++] = RowPart{TALL, 0, full - 256, full / 256 - 1};
    out[np++] = RowPart{(rest <= 4 && allow_mini) ? TALL_MINI : TALL_STRIP, full - 256, 256 + rest, 1};
  } else if (rest <= 64) {
    out[np++] = RowPart{TALL, 0, full, full / 256};
    out[np++] = RowPart{rest <= 32 ? WIDE32 : WIDE, full, rest, 1};
  } else if (last >= 0 && extended) {
    // 256 a + 65 ... 96 / 129 ... 192 rows: instead of three 128-row blocks for two and a bit, or a 256-row block that is half to
    // three quarters full — the 256-row geometry for the full blocks, the rest in its own height (a second pass over the panel)
    out[np++] = RowPart{TALL, 0, full, full / 256};
    out[np++] = RowPart{last, full, rest, 1};
  }
  return np;
}

// rows of a tile of each geometry (the strip: 256 + 16)
inline uint32_t tile_rows(int geo) {
  static const uint32_t bm[GEO_COUNT] = {64, 128, 256, 64, 32, 272, 96, 192, 160, 224, 272, 144, 80};
  return bm[geo];
}

struct GeoModel {
  int bn;            // columns per tile
  int per_cu;        // resident workgroups per CU
  double us[5];      // one round: microseconds per k chunk of a tile with 1 .. per_cu workgroups sharing the CU
  double stream_us;  // several rounds: microseconds per k chunk and tile of a CU's share (workgroups out of phase overlap their tile ends)
  double tile_us;    // ... plus this much per tile (N = 128 against N = 256: the tile ends do not shrink with k)
};

inline const GeoModel& model(int geo) {
  static const GeoModel table[GEO_COUNT] = {
      /* FLAT   64 x 64  */ {64, 5, {0.65, 1.10, 1.80, 2.80, 2.90}, 0.48, 0.6},
      /* MID   128 x 128 */ {128, 2, {2.00, 4.00, 0, 0, 0}, 2.00, 0.0},
      /* TALL  256 x 128 */ {128, 1, {3.90, 0, 0, 0, 0}, 3.90, 0.0},
      /* WIDE   64 x 256 */ {256, 2, {2.10, 3.90, 0, 0, 0}, 1.95, 0.0},
      /* WIDE32 32 x 256 */ {256, 2, {1.30, 2.30, 0, 0, 0}, 1.15, 0.0},
      /* STRIP 272 x 128 */ {128, 1, {4.30, 0, 0, 0, 0}, 4.30, 0.0},
      /* G96    96 x 128 */ {128, 2, {2.00, 3.80, 0, 0, 0}, 1.90, 0.0},    // (N = 96: 10.7 rounds in 244.7 us — the memory system's pace, not the MFMA pipe's)
      /* G192  192 x 64  */ {64, 2, {1.70, 3.20, 0, 0, 0}, 1.60, 0.0},     // (N = 192: 10.7 rounds in 408.5 us)
      /* G160  160 x 64  */ {64, 2, {1.50, 2.80, 0, 0, 0}, 1.40, 0.0},     // (N = 160: 12.8 rounds in 355.6 us)
      /* G224  224 x 64  */ {64, 2, {2.00, 3.75, 0, 0, 0}, 1.90, 0.0},     // (N = 224: 9.14 rounds in 480.7 us)
      /* MINI  256+4 x 128 */ {128, 1, {4.15, 0, 0, 0, 0}, 4.15, 0.0},    // (the tall geometry + four 17-cycle MFMAs per stage and wave; N = 257: 8 rounds x 17 chunks in 565 us)
      /* MMINI 128+4 x 128 */ {128, 2, {2.10, 4.20, 0, 0, 0}, 2.10, 0.0},    // (the 128-row geometry + eight 17-cycle MFMAs per stage and wave)
      /* FMINI  64+4 x 64  */ {64, 4, {0.70, 1.20, 1.95, 2.95, 0}, 0.52, 0.6},  // (the 64-row geometry + four 17-cycle MFMAs per stage and wave)
  };
  return table[geo];
}

constexpr double kLaunchOverheadUs = 3.5;

struct Launch {
  int geo = TALL;
  uint32_t row_blocks = 1;
  int per_cu = 0;            // resident workgroups per CU as MEASURED for this device by prepare() (0: the table's figure)
};

// microseconds per k chunk of a tile when `sharing` workgroups share the CU (1-based); beyond what the table was measured for —
// a device that keeps more workgroups resident than the MI355X the table comes from — the last figure grows in proportion
inline double shared_us(const GeoModel& g, int sharing) {
  int last = 0;
  for (int i = 0; i < 5; ++i)
    if (g.us[i] > 0.0) last = i;
  if (sharing - 1 <= last) return g.us[sharing < 1 ? 0 : sharing - 1];
  return g.us[last] * static_cast<double>(sharing) / static_cast<double>(last + 1);
}

// one persistent launch of `l` over `cols` columns (0 columns: no launch)
inline double launch_us(const Launch& l, size_t cols, uint32_t nk, int cus) {
  if (cols == 0) return 0.0;
  const GeoModel& g = model(l.geo);
  const int per_cu = l.per_cu > 0 ? l.per_cu : g.per_cu;       // (what THIS device keeps resident, when the caller knows)
  const double tiles = std::ceil(static_cast<double>(cols) / g.bn) * l.row_blocks;
  const double slots = static_cast<double>(cus) * per_cu;
  double t, per_tile = 0.0;
  if (tiles <= slots) {
    // one round: the grid is the tiles and the hardware spreads it over the CUs
    t = shared_us(g, static_cast<int>(std::ceil(tiles / cus)));
  } else if (per_cu > 2) {
    // many small workgroups per CU: the CU works through its share of the tiles like a queue (measured: 3 072 tiles of 64 x 64
    // take 101 us where 3 rounds of 5 per CU would take 130)
    t = std::ceil(tiles / cus) * g.stream_us;
    per_tile = std::ceil(tiles / cus) * g.tile_us;
    // ... and every row block re-reads the panel: 64 columns x K x 8 bytes in and 64 x 64 x 8 out per tile, at what the memory
    // system gives a launch whose panel no longer fits the 256 MB Infinity Cache (measured: N = 320, 209 664 columns in 64 x 64
    // tiles alone move 3.2 GB in 756 us) — above four row blocks this, not the MFMA pipe, is what bounds the small tiles
    const double k_bytes = 16.0 * nk * 8.0;
    const double bytes = tiles * (g.bn * k_bytes + g.bn * 64.0 * 8.0);
    const double panel = static_cast<double>(cols) * k_bytes;
    const double bytes_per_us = (panel <= 128e6 ? 5.2e6 : 4.3e6) * static_cast<double>(cus) / 256.0;   // (measured on 256 CUs / 8 HBM stacks)
    if (bytes / bytes_per_us > t * nk + per_tile) return bytes / bytes_per_us + kLaunchOverheadUs;
  } else {
    // whole rounds; the tiles behind the last full round go to the FIRST workgroups of a grid that sits per_cu to a CU: up to one
    // per CU they may or may not share (measured: 768 tiles of 128 x 128 take 125 us where two full rounds take 134 and
    // "one round + one tile alone" 101; 768 tiles of 64 x 256 behave like the latter) — the mean of both
    const double full = std::floor(tiles / slots);
    const double rem = tiles - full * slots;
    t = full * shared_us(g, per_cu);
    if (rem > 0.0) {
      if (per_cu == 2 && rem <= cus / 2) t += g.us[0];                                 // (a few tiles: each on a CU of its own)
      else if (per_cu == 2 && rem <= cus) t += 0.5 * (g.us[0] + g.us[1]);
      else t += shared_us(g, per_cu);
    }
  }
  return t * nk + per_tile + kLaunchOverheadUs;
}

struct Choice {
  size_t primary_cols = 0;   // columns [0, primary_cols) go to the operator's own launches ...
  int alt = -1;              // ... the rest to alternates[alt] (-1: there is no rest)
  double model_us = 0.0;     // what the model expects of this choice
  double primary_only_us = 0.0;
};

// `primary`: the launches the operator's own path makes over its columns (one, or two for an operator in parts);
// `alternates`: finer geometries of the WHOLE operator.  `may_split` false: one geometry for all columns (the fused step, whose
// convergence monitors belong to the workgroups of ONE grid) — the choice is primary_cols == M or primary_cols == 0.
inline Choice choose(const Launch* primary, int num_primary, const Launch* alternates, int num_alternates, size_t M, uint32_t nk, int cus,
                     bool may_split = true) {
  auto primary_us = [&](size_t cols) {
    double t = 0.0;
    for (int i = 0; i < num_primary; ++i) t += launch_us(primary[i], cols, nk, cus);
    return t;
  };
  Choice best;
  best.primary_cols = M;
  best.alt = -1;
  best.model_us = best.primary_only_us = primary_us(M);
  if (M == 0 || num_alternates == 0) return best;
  const GeoModel& g0 = model(primary[0].geo);
  // columns of a full round of the first (dominant) primary launch, in whole tiles of the widest primary tile (256)
  size_t round_cols = static_cast<size_t>(cus) * (primary[0].per_cu > 0 ? primary[0].per_cu : g0.per_cu) / primary[0].row_blocks * g0.bn;
  round_cols -= round_cols % 256;
  Choice cand = best;
  for (int a = 0; a < num_alternates; ++a) {
    size_t keeps[2] = {0, (may_split && round_cols != 0) ? M / round_cols * round_cols : 0};
    for (size_t keep : keeps) {
      if (keep >= M) continue;
      const double t = primary_us(keep) + launch_us(alternates[a], M - keep, nk, cus);
      if (t < cand.model_us) {
        cand.primary_cols = keep;
        cand.alt = a;
        cand.model_us = t;
      }
    }
  }
  // (short k: the tile ends, which the table folds into its per-chunk figures at K = 256, weigh more — ask for more)
  const double needed = nk >= 16 ? 0.95 : 0.90;
  if (cand.alt >= 0 && cand.model_us <= needed * best.model_us && cand.model_us + 3.0 <= best.model_us) return cand;
  return best;
}

}  // namespace plan
}  // namespace sri

#endif  // SRI_COLUMN_PLAN_H
