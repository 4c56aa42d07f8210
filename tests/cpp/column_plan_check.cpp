// column_plan_check.cpp — csrc/sri_column_plan.h on the CPU: the choices it makes at the sizes that were measured on the
// MI355X (profiles/r04_geometry_by_columns.log), and the invariants run() relies on for every size.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "sri_column_plan.h"

using namespace sri::plan;

static int failures = 0;
#define CHECK(cond, ...)                                      \
  do {                                                        \
    if (!(cond)) {                                            \
      ++failures;                                             \
      std::printf("FAILED %s:%d: %s  ", __FILE__, __LINE__, #cond); \
      std::printf(__VA_ARGS__);                               \
      std::printf("\n");                                      \
    }                                                         \
  } while (0)

int main() {
  const int cus = 256;
  // a residency the table has no figure for grows in proportion; the measured one, when given, is what counts
  CHECK(shared_us(model(TALL), 1) == 3.90 && shared_us(model(TALL), 2) == 7.80, "tall, two to a CU: %.2f", shared_us(model(TALL), 2));
  CHECK(shared_us(model(FLAT), 5) == 2.90 && shared_us(model(FLAT), 6) > 2.90, "flat, six to a CU: %.2f", shared_us(model(FLAT), 6));
  {
    Launch flat3{FLAT, 4, 3}, flat5{FLAT, 4, 0};
    CHECK(launch_us(flat3, 4096, 16, 256) == launch_us(flat5, 4096, 16, 256), "a single round of one tile per CU does not depend on the residency");
    CHECK(launch_us(flat3, 49152, 16, 256) > 0.0, "three to a CU");
  }
  // ---- N = 256 (one 256-row block, 16 chunks): alternates 128 x 128 (2 row blocks) and 64 x 64 (4 row blocks)
  const Launch tall{TALL, 1};
  const Launch finer[2] = {{MID, 2}, {FLAT, 4}};
  struct Expect { size_t cols; size_t kept; int alt; };
  const Expect table[] = {
      {4096, 0, 1},          // 32 tall tiles: measured 61.7 / 34.0 / 13.2 us -> all in 64 x 64 tiles
      {8192, 0, 1},          // 61.9 / 35.4 / 20.3
      {16384, 0, 0},         // 64.2 / 38.6 / 49.5 -> 128 x 128
      {32768, 32768, -1},    // one full round: 70.8 / 69.2 / 73.7 -> stays
      {262144, 262144, -1},  // the BASELINE panel: 8 full rounds
      {49152, 32768, 0},     // 1.5 rounds: one round tall + the rest finer (measured whole: 128.5 / 124.7 / 101.4)
  };
  for (const Expect& e : table) {
    const Choice c = choose(&tall, 1, finer, 2, e.cols, 16, cus);
    CHECK(c.primary_cols == e.kept && c.alt == e.alt, "cols %zu: kept %zu alt %d (model %.1f us, primary only %.1f us)", e.cols,
          c.primary_cols, c.alt, c.model_us, c.primary_only_us);
  }
  // the model against the measurements it was fitted to (within 15 %)
  struct Timed { Launch l; size_t cols; double us; };
  const Timed timed[] = {{{TALL, 1}, 4096, 61.7},  {{MID, 2}, 4096, 34.0},   {{FLAT, 4}, 4096, 13.2},  {{FLAT, 4}, 8192, 20.3},
                         {{MID, 2}, 16384, 38.6},  {{FLAT, 4}, 16384, 49.5}, {{TALL, 1}, 32768, 70.8}, {{MID, 2}, 32768, 69.2},
                         {{TALL, 1}, 49152, 128.5}, {{FLAT, 4}, 49152, 101.4}, {{TALL, 1}, 262144, 498.2}};
  for (const Timed& t : timed) {
    const double m = launch_us(t.l, t.cols, 16, cus);
    CHECK(m > 0.85 * t.us && m < 1.15 * t.us, "geo %d cols %zu: model %.1f us, measured %.1f us", t.l.geo, t.cols, m, t.us);
  }
  // ---- N = 320 in parts (256-row main part + 64 x 256 last part, 20 chunks), alternates of the whole operator
  {
    const Launch own[2] = {{TALL, 1}, {WIDE, 1}};
    const Launch alt[2] = {{MID, 3}, {FLAT, 5}};
    const double whole = launch_us(own[0], 209664, 20, cus) + launch_us(own[1], 209664, 20, cus);
    CHECK(whole > 0.9 * 702.7 && whole < 1.1 * 702.7, "N = 320 sweep panel: model %.1f us, measured 702.7 us", whole);
    const Choice c = choose(own, 2, alt, 2, 209664, 20, cus);
    CHECK(c.alt < 0 || (c.primary_cols % 256 == 0 && c.primary_cols < 209664), "kept %zu", c.primary_cols);
  }
  // ---- invariants, random shapes
  std::mt19937_64 rng(12345);
  for (int it = 0; it < 200000; ++it) {
    const uint32_t rows = 65 + rng() % 2000;
    const uint32_t nk = (rows + 15) / 16;
    Launch own[2];
    int num_own = 1;
    const int base = rng() % 2 ? TALL : MID;
    const uint32_t bm = base == TALL ? 256 : 128;
    own[0] = Launch{base, (rows + bm - 1) / bm};
    if (rng() % 3 == 0) own[num_own++] = Launch{static_cast<int>(rng() % 2 ? WIDE : WIDE32), 1};
    Launch alt[2];
    int num_alt = 0;
    if (base == TALL) alt[num_alt++] = Launch{MID, (rows + 127) / 128};
    if (rng() % 4 != 0) alt[num_alt++] = Launch{FLAT, (rows + 63) / 64};
    const size_t M = 1 + rng() % (rng() % 2 ? 5000 : 600000);
    // other devices: fewer CUs (a partitioned MI355X shows 32 or 64), another residency than the table's (prepare() measures it)
    static const int cu_counts[4] = {256, 256, 64, 32};
    const int cus = cu_counts[rng() % 4];
    if (rng() % 3 == 0) {
      own[0].per_cu = 1 + static_cast<int>(rng() % 2);
      for (int a = 0; a < num_alt; ++a) alt[a].per_cu = 1 + static_cast<int>(rng() % 6);
    }
    const Choice c = choose(own, num_own, alt, num_alt, M, nk, cus);
    CHECK(c.primary_cols <= M, "kept %zu of %zu", c.primary_cols, M);
    CHECK((c.alt < 0) == (c.primary_cols == M), "alt %d kept %zu of %zu", c.alt, c.primary_cols, M);
    CHECK(c.alt < num_alt, "alt %d of %d", c.alt, num_alt);
    CHECK(c.primary_cols == M || c.primary_cols % 256 == 0, "kept %zu: whole tiles of every primary geometry", c.primary_cols);
    CHECK(c.model_us <= c.primary_only_us, "model %.2f > primary only %.2f", c.model_us, c.primary_only_us);
    if (c.alt >= 0) CHECK(c.model_us <= (nk >= 16 ? 0.95 : 0.90) * c.primary_only_us, "deviates for %.2f against %.2f", c.model_us, c.primary_only_us);
    const Choice none = choose(own, num_own, alt, 0, M, nk, cus);
    CHECK(none.alt < 0 && none.primary_cols == M, "no alternates: nothing to choose");
  }
  // ---- rows: own geometry and parts
  {
    struct ExpectRows { uint32_t rows; int own; int np; int g0, g1; };
    const ExpectRows rows_table[] = {
        {16, FLAT, 0, -1, -1},   {64, FLAT, 0, -1, -1},      {65, MID, 1, FLAT_MINI, -1},   {68, MID, 1, FLAT_MINI, -1},   {69, MID, 1, G96, -1},        {96, MID, 1, G96, -1},      {97, MID, 0, -1, -1},      {128, MID, 0, -1, -1},
        {129, TALL, 1, MID_MINI, -1}, {132, TALL, 1, MID_MINI, -1}, {133, TALL, 1, G160, -1},     {160, TALL, 1, G160, -1}, {161, TALL, 1, G192, -1},   {192, TALL, 1, G192, -1},
        {193, TALL, 1, G224, -1}, {224, TALL, 1, G224, -1}, {225, TALL, 0, -1, -1}, {255, TALL, 0, -1, -1},     {256, TALL, 0, -1, -1},       {257, MID, 1, TALL_MINI, -1},
        {260, MID, 1, TALL_MINI, -1}, {261, MID, 1, TALL_STRIP, -1}, {272, MID, 1, TALL_STRIP, -1}, {273, MID, 2, TALL, WIDE32}, {288, MID, 2, TALL, WIDE32}, {289, MID, 2, TALL, WIDE},
        {320, MID, 2, TALL, WIDE}, {321, MID, 2, TALL, FLAT_MINI}, {325, MID, 2, TALL, G96},  {352, MID, 2, TALL, G96},   {353, MID, 0, -1, -1},   {384, MID, 0, -1, -1},        {385, TALL, 2, TALL, MID_MINI}, {388, TALL, 2, TALL, MID_MINI}, {389, TALL, 2, TALL, G160},
        {416, TALL, 2, TALL, G160}, {417, TALL, 2, TALL, G192}, {448, TALL, 2, TALL, G192}, {449, TALL, 0, -1, -1},    {480, TALL, 0, -1, -1},
        {512, TALL, 0, -1, -1},  {513, MID, 2, TALL, TALL_MINI}, {517, MID, 2, TALL, TALL_STRIP}, {528, MID, 2, TALL, TALL_STRIP}, {529, MID, 2, TALL, WIDE32},
        {576, MID, 2, TALL, WIDE}, {600, MID, 2, TALL, G96},  {640, MID, 0, -1, -1},     {700, TALL, 2, TALL, G192},     {1024, TALL, 0, -1, -1},      {1025, MID, 2, TALL, TALL_MINI},
        {1088, MID, 2, TALL, WIDE},
    };
    for (const ExpectRows& e : rows_table) {
      RowPart rp[2];
      const int np = row_parts(e.rows, true, rp);
      CHECK(own_geometry(e.rows) == e.own, "rows %u: own geometry %d", e.rows, own_geometry(e.rows));
      CHECK(np == e.np && (np < 1 || rp[0].geo == e.g0) && (np < 2 || rp[1].geo == e.g1), "rows %u: %d parts (%d, %d)", e.rows, np,
            np > 0 ? rp[0].geo : -1, np > 1 ? rp[1].geo : -1);
    }
    for (uint32_t rows = 1; rows <= 4200; ++rows) {
      for (int strip = 0; strip < 2; ++strip) {
        RowPart rp[2];
        const int np = row_parts(rows, strip != 0, rp);
        CHECK(np >= 0 && np <= 2, "rows %u: %d parts", rows, np);
        if (np == 0) continue;
        // the parts tile [0, rows) without gap or overlap, each starts on a row block of its own geometry, none is empty
        uint32_t next = 0, padded = 0;
        for (int i = 0; i < np; ++i) {
          CHECK(rp[i].row0 == next && rp[i].rows > 0, "rows %u part %d: row0 %u rows %u", rows, i, rp[i].row0, rp[i].rows);
          const uint32_t bm = tile_rows(rp[i].geo);
          CHECK(rp[i].rows <= rp[i].row_blocks * bm && rp[i].rows > (rp[i].row_blocks - 1) * bm, "rows %u part %d: %u rows in %u blocks of %u",
                rows, i, rp[i].rows, rp[i].row_blocks, bm);
          CHECK(rp[i].row0 % 64 == 0, "rows %u part %d starts at %u", rows, i, rp[i].row0);
          CHECK(rp[i].geo != TALL_STRIP || (strip != 0 && rp[i].row_blocks == 1 && rp[i].rows > 256 && rp[i].rows <= 272), "rows %u: strip of %u rows",
                rows, rp[i].rows);
          CHECK(rp[i].geo != TALL_MINI || (strip != 0 && rp[i].row_blocks == 1 && rp[i].rows > 256 && rp[i].rows <= 260), "rows %u: mini strip of %u rows",
                rows, rp[i].rows);
          CHECK(rp[i].geo != FLAT_MINI || (strip != 0 && rp[i].row_blocks == 1 && rp[i].rows > 64 && rp[i].rows <= 68), "rows %u: flat mini strip of %u rows",
                rows, rp[i].rows);
          CHECK(rp[i].geo != MID_MINI || (strip != 0 && rp[i].row_blocks == 1 && rp[i].rows > 128 && rp[i].rows <= 132), "rows %u: mid mini strip of %u rows",
                rows, rp[i].rows);
          CHECK(i == 0 || rp[i].geo == TALL_STRIP || rp[i].geo == TALL_MINI || rp[i].rows <= 192, "rows %u: last part of %u rows", rows, rp[i].rows);
          next += rp[i].rows;
          padded += rp[i].row_blocks * ((rp[i].geo == TALL_STRIP || rp[i].geo == TALL_MINI) ? 256 + 16 * ((rp[i].rows - 256 + 15) / 16) : bm);
        }
        CHECK(next == rows, "rows %u: parts cover %u", rows, next);
        // what parts are for: never more padding than the operator kept whole in its own geometry, and at most 63 padded rows
        const uint32_t own_bm = tile_rows(own_geometry(rows));
        const uint32_t whole = (rows + own_bm - 1) / own_bm * own_bm;
        CHECK(padded <= whole && padded - rows < 64, "rows %u: %u padded rows in parts, %u whole", rows, padded, whole);
      }
    }
  }
  std::printf("column_plan_check: %d failures\n", failures);
  return failures == 0 ? 0 : 1;
}
