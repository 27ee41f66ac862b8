// orbx_knobs.h — diagnostic knobs of liborbx (orbx_debug_set, include/orbx.h).  The shipped library reads no environment
// variable: tests, tools and experiments that want a particular kernel or launch shape set a named integer knob through the C ABI
// (the Python loader forwards ORBX_<NAME> environment variables to it, so the profiling scripts keep their command lines).
// None of the knobs changes a result: they choose among kernels / launch shapes that produce identical output.
#pragma once
#include <atomic>
#include <climits>

namespace orbx {

// X(enumerator, "name")
#define ORBX_KNOB_LIST(X)                                                                                                        \
  X(KNOB_NO_BANDS, "no_bands")                /* pyramid: never k_pyramid_bands */                                               \
  X(KNOB_NO_TILES, "no_tiles")                /* pyramid: small batches launch the levels one by one */                          \
  X(KNOB_TILES_MAX_FRAMES, "tiles_max_frames") /* k_pyramid_tiles up to this many frames per launch (default 8) */               \
  X(KNOB_TILES_MAX_PIXELS, "tiles_max_pixels") /* ... and this many pixels per launch */                                         \
  X(KNOB_PYR_BANDS, "pyr_bands")              /* bands per frame of k_pyramid_bands */                                           \
  X(KNOB_PYR_GMAX, "pyr_gmax")                /* k_pyramid_bands: 1 = one group of 4 pixels per thread, 2 = up to two */          \
  X(KNOB_PYR_STRIPS, "pyr_strips")            /* column strips per band of k_pyramid_bands */                                    \
  X(KNOB_BANDS_MIN_FRAMES, "bands_min_frames") /* k_pyramid_bands from this many frames per stream */                            \
  X(KNOB_DESC_NO_STAGED, "desc_no_staged")    /* small launches keep k_sel_compact */                                            \
  X(KNOB_DESC_STAGED_MAX, "desc_staged_max")  /* units up to which k_describe_patch reads the staging lists itself */            \
  X(KNOB_NO_SPLIT, "no_split")                /* synchronous calls: one stream instead of two half batches */                    \
  X(KNOB_LAT_TRACE, "lat_trace")              /* orbx_extract prints a host-side timeline to stderr */                           \
  X(KNOB_NO_DIRECT_OUT, "no_direct_out")      /* the one-frame call copies its results through the staging buffers */            \
  X(KNOB_FAST_WG, "fast_wg")                  /* always k_fast (a workgroup per cell) */                                         \
  X(KNOB_FAST_WG_MAX_CELLS, "fast_wg_max_cells") /* launches of up to this many cells take k_fast (default 5000) */              \
  X(KNOB_FAST_LDS_PAD, "fast_lds_pad")        /* unused dynamic LDS per k_fast_wave workgroup: fewer waves per CU */             \
  X(KNOB_FAST_DEBUG, "fast_debug")            /* print k_fast_wave's occupancy */                                                \
  X(KNOB_DESC_LDS_PAD, "desc_lds_pad")        /* unused dynamic LDS per k_describe_patch workgroup */                            \
  X(KNOB_MATCH_NO_GENERAL, "match_no_general") /* the matcher's sequential fallback is skipped (pairs it would take stay pending) */ \
  X(KNOB_MATCH_NO_MFMA, "match_no_mfma")      /* brute-force matching on the vector ALU (xor / bcnt) instead of the matrix cores */ \
  X(KNOB_OCT_NO_SMALL, "oct_no_small")        /* selection: always the 2048-candidate LDS instance */                            \
  X(KNOB_OCT_KEY64, "oct_key64")              /* selection: always 64-bit sort keys */                                           \
  X(KNOB_OCT_SPLIT_MIN, "oct_split_min")      /* batch size from which every group of levels with one instance gets its own launch */ \
  X(KNOB_OCT_NO_BIG, "oct_no_big")            /* large units on one workgroup (k_octree_global) instead of the bucket kernels */ \
  X(KNOB_OCT_BIG_DEPTH, "oct_big_depth")      /* bucket depth of the many-workgroup selection (tests: a depth whose buckets overflow) */ \
  X(KNOB_OCT_BIG_NO_FALLBACK, "oct_big_no_fallback") /* k_octree_big does not redo a unit it cannot take: the unit fails (tests) */ \
  X(KNOB_OCTB_NO_512, "octb_no_512")          /* k_octree_buckets: always 1024 LDS slots per wave */                             \
  X(KNOB_OCT_INST, "oct_inst")                /* LDS instance per level, one hex digit per level from level 0 upwards (lowest digit
                                                 first): 1 = 512, 2 = 1024, 3 = 2048, 0 = the default choice */                 \
  X(KNOB_OCT_LDS_PAD, "oct_lds_pad")          /* unused dynamic LDS per selection workgroup: fewer units per CU */               \
  X(KNOB_MULTI_FORCE_RCCL, "multi_force_rccl") /* orbx_multi_create: a one-device context goes through RCCL as well */

enum Knob {
#define ORBX_KNOB_ENUM(e, n) e,
  ORBX_KNOB_LIST(ORBX_KNOB_ENUM)
#undef ORBX_KNOB_ENUM
  KNOB_COUNT
};
constexpr long long KNOB_UNSET = LLONG_MIN;
extern std::atomic<long long> g_knob[KNOB_COUNT];  // (orbx_api.cpp)

inline long long knob(Knob k, long long dflt) {
  const long long v = g_knob[k].load(std::memory_order_relaxed);
  return v == KNOB_UNSET ? dflt : v;
}
inline bool knobOn(Knob k) { return knob(k, 0) != 0; }

}  // namespace orbx
