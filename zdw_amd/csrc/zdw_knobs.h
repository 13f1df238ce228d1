// zdw_knobs.h — the environment switches of libzdw_hip.so, in one place.
//
// Two kinds, and nothing else in the library calls getenv:
//   test_hook(name)    forces one of the forms a kernel or a host path has (tests/ run every reference vector through each form:
//                      ZDW_P1_SKIP_EMPTY, ZDW_P1_NO_FOLD, ZDW_P1_ALL_IN_ONE, ZDW_P1_NO_NZ, ZDW_P1_REBASE_AT, ZDW_SPAN_SHIFT,
//                      ZDW_P2_DENSE, ZDW_P2_DIRECT, ZDW_P2_COLS_MIN, ZDW_P2_EMIT_ROWS, ZDW_P2_EMIT_NOPIPE, ZDW_P2_LOOP, ZDW_P2_LOOP_STAGE, ZDW_STREAM_PIECE,
//                      ZDW_STREAM_CAP0, ZDW_DEC_GROUP, ZDW_P1_LARGE_RING, ZDW_P1_TRACK_FIRST, ZDW_MODEL_RERUN, ZDW_DICT_SLOTS, ZDW_P1_WPB,
//                      ZDW_SORT_NO_FINISH, ZDW_SORT_SCATTER, ZDW_SORT_NO_PACK, ZDW_SORT_PACK, ZDW_TRIM_BY_SCAN, ZDW_P1_ROWS (pass 1 with a lane per row, csrc/k_rows.hip: 0 never -- the default --, 1 where it is
//                      expected to pay, 2 wherever it can run), ZDW_P1_ROWS_PER_LANE, ZDW_P1_ROWS_WPB).  Every
//                      form writes the same bytes; the switches exist in the shipped build so that the shipped code is what is tested.
//   experiment(name)   tuning and A/B measurements of the profiling tools (grids, queue sizes, transports, debug output): compiled
//                      in only with -DZDW_EXPERIMENTS (tools/build_experiments.sh); the shipped build ignores them.
#pragma once
#include <cstdlib>

namespace zdw {

inline const char *test_hook(const char *name) { return getenv(name); }

inline const char *experiment(const char *name)
{
#ifdef ZDW_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace zdw
