#pragma once
// resblock_launch.h — part 4 of 4: persistent grids, run shares of the dispatch classes, launchers.
#include "resblock_kernel.h"

namespace {

// Offline carry form with several workgroups per CU: runs in proportion to the speeds of the dispatch classes (see resblock_kernel)
inline void set_class_shares(ResArgs& a, long blocks, long resident, int n_cu) {
  a.classes = 0;
  if (n_cu > 0 && blocks == resident && resident % n_cu == 0 && resident / n_cu >= 2 && resident / n_cu <= 4 &&
      a.total_tiles >= 8 * resident) {
    const int cls = (int)(resident / n_cu);
    // shares of the dispatch classes (first-dispatched first), measured: see tools/res_wg_times.py and profiles/r03_experiments.md
    double share[4] = {0, 0, 0, 0};
    if (cls == 2) { share[0] = HILC_RES_SHARE2_0; share[1] = 1.0 - share[0]; }
    else if (cls == 3) { share[0] = HILC_RES_SHARE3_0; share[1] = HILC_RES_SHARE3_1; share[2] = 1.0 - share[0] - share[1]; }
    else { for (int i = 0; i < cls; ++i) share[i] = 1.0 / cls; }
#ifdef HILC_RES_SHARE_ENV      // tuning builds only
    if (cls == 2) { if (const char* e = getenv("HILC_SHARE2_0")) { share[0] = atof(e); share[1] = 1.0 - share[0]; } }
    if (cls == 3) {
      if (const char* e = getenv("HILC_SHARE3_0")) share[0] = atof(e);
      if (const char* e = getenv("HILC_SHARE3_1")) share[1] = atof(e);
      share[2] = 1.0 - share[0] - share[1];
    }
#endif
    double acc = 0;
    a.cum[0] = 0;
    for (int i = 0; i < cls; ++i) { acc += share[i]; a.cum[i + 1] = (unsigned)(acc * 65536.0 + 0.5); }
    a.cum[cls] = 65536u;
    a.classes = cls;
  }
}

// number of workgroups of this instantiation that can be resident on the device (CUs x occupancy), cached per device
template <class KernelT>
int resident_workgroups(KernelT kernel, int threads, std::atomic<int>* cache, int& n_cu_out) {
  constexpr int MAXDEV = 64;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  int n_cu = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) return -1;
  n_cu_out = n_cu;
  int cached = dev >= 0 && dev < MAXDEV ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (cached == 0) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, 0) != hipSuccess || occ < 1) return -1;
    cached = n_cu * occ;
    if (dev >= 0 && dev < MAXDEV) cache[dev].store(cached, std::memory_order_relaxed);
  }
  return cached;
}

inline void set_div_magic(ResArgs& a) {
  // division by the invariant T (Granlund-Montgomery, 31-bit dividends): l = ceil(log2 T), m = ceil(2^(31+l) / T)
  int l = 0;
  while ((1L << l) < a.T) ++l;
  if (l < 1) l = 1;
  const unsigned long long p = 1ULL << (31 + l);
  a.div_magic = (unsigned)((p + (unsigned long long)a.T - 1) / (unsigned long long)a.T);
  a.div_shift = (unsigned)(l - 1);
}

// A CHAIN launch: a.nblk blocks of one stage, one launch.  Offline: the carry form's contiguous runs (equal split).  STREAM,
// C <= 192: runs of whole streams on the flat column space — the shortest run is `unit` tiles = the fewest whole streams that
// fill whole tiles; runs are made of as many units as it takes for all runs to be resident at once (1024 streams x 320 samples,
// 256 workgroups: 4 streams = 10 tiles each), the last run may be short.  STREAM NARROW (C >= 512): whole-stream tiles, static
// stride.  Returns HILC_ERR_UNSUPPORTED where the geometry does not fit (the caller launches the blocks one by one).
template <int C, bool STREAM, int NB, bool W8, int DR = 0, bool POST = false, bool SPEC0 = false>      // DR > 0: + down-sampling phase, DR < 0: + up-sampling phase (r = -DR); POST: + closing conv; SPEC0: + stage-0 input phase
int launch_chain(ResArgs a, int B, hipStream_t s) {
  constexpr bool SC = STREAM && C <= 384;         // runs of whole streams with carries (C = 256 / 384: 32-column tiles; C >= 512: whole-stream tiles, static stride)
  using K = Cfg<C, STREAM, SC, NB, W8, DR, POST, SPEC0>;
  a.B = B;
  set_div_magic(a);
  constexpr int TO = K::TO;
  a.tiles = (a.T + TO - 1) / TO;
  a.total_tiles = STREAM ? ((long)B * a.T + TO - 1) / TO : (long)B * a.tiles;
  a.classes = 0;
  a.run_tiles = 0;
  static std::atomic<int> resident_cache[64];
  int n_cu = 0;
  const long resident = resident_workgroups(resblock_kernel<C, STREAM, SC, NB, W8, DR, POST, SPEC0>, K::NT, resident_cache, n_cu);
  if (resident < 1) return HILC_ERR_LAUNCH;
  long blocks;
  if constexpr (SC) {
    long g = a.T, h = K::NCOL;
    while (h != 0) { const long t = g % h; g = h; h = t; }                      // gcd(T, NCOL)
    const long unit = (long)a.T / g;                                              // tiles of the shortest run of whole streams
    const long units = (a.total_tiles + unit - 1) / unit;
    const long k = (units + resident - 1) / resident;                            // units per run
    a.run_tiles = k * unit;
    blocks = (a.total_tiles + a.run_tiles - 1) / a.run_tiles;
  } else {
    blocks = a.total_tiles < resident ? a.total_tiles : resident;
    if constexpr (!STREAM) set_class_shares(a, blocks, resident, n_cu);
  }
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL((resblock_kernel<C, STREAM, SC, NB, W8, DR, POST, SPEC0>), dim3((unsigned)blocks), dim3(K::NT), 0, s, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

template <int C, bool STREAM, bool SCARRY = false>
int launch_res(ResArgs a, int B, hipStream_t s, long carry_grid = 0) {
  a.nblk = 1;
  a.run_tiles = 0;
  a.B = B;
  {  // division by the invariant T (Granlund-Montgomery, 31-bit dividends): l = ceil(log2 T), m = ceil(2^(31+l) / T)
    int l = 0;
    while ((1L << l) < a.T) ++l;
    if (l < 1) l = 1;
    const unsigned long long p = 1ULL << (31 + l);
    a.div_magic = (unsigned)((p + (unsigned long long)a.T - 1) / (unsigned long long)a.T);
    a.div_shift = (unsigned)(l - 1);
  }
  using K = Cfg<C, STREAM, SCARRY>;
  constexpr int TO = K::TO;
  a.tiles = (a.T + TO - 1) / TO;
  a.total_tiles = STREAM ? ((long)B * a.T + TO - 1) / TO : (long)B * a.tiles;
  // persistent grid = exactly what can be resident (a surplus workgroup would only start after a resident one has
  // walked its whole tile list).  Immutable per-device facts, looked up once per device (a process may drive several GPUs).
  constexpr int MAXDEV = 64;
  static std::atomic<int> resident_cache[MAXDEV];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return HILC_ERR_LAUNCH;
  int cached = dev >= 0 && dev < MAXDEV ? resident_cache[dev].load(std::memory_order_relaxed) : 0;
  if (cached == 0) {
    int n_cu = 0, occ = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1)
      return HILC_ERR_LAUNCH;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, resblock_kernel<C, STREAM, SCARRY>, K::NT, 0) != hipSuccess || occ < 1)
      return HILC_ERR_LAUNCH;
    cached = n_cu * occ;
    if (dev >= 0 && dev < MAXDEV) resident_cache[dev].store(cached, std::memory_order_relaxed);
  }
  const long resident = cached;
  if constexpr (STREAM && !SCARRY && (C == 96 || C == 192)) {
    // The carry form for this hop?  Only where every run is the same whole number of streams (so that no run starts inside a
    // stream and pays a warm-up tile) and the runs are fewer tile-times than the rounds of the halo form.
    constexpr int NC = K::NCOL;
    long g = a.T, h = NC;
    while (h != 0) { const long t = g % h; g = h; h = t; }                      // gcd(T, NCOL)
    const long unit = (long)a.T / g;                                              // tiles of the shortest aligned run
    const long cols = (long)B * a.T;
    if (cols % (unit * NC) == 0) {
      const long units = cols / (unit * NC);
      const long k = (units + resident - 1) / resident;                          // aligned runs per workgroup (same residency: 8 KB of LDS more)
      const long halo_rounds = (a.total_tiles + resident - 1) / resident;
      if (units % k == 0 && k * unit < halo_rounds) return launch_res<C, STREAM, true>(a, B, s, units / k);
    }
  }
  long blocks = a.total_tiles < resident ? a.total_tiles : resident;
  if (SCARRY && carry_grid > 0 && carry_grid <= resident) blocks = carry_grid;
  a.classes = 0;
  if constexpr (!STREAM) {
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) set_class_shares(a, blocks, resident, n_cu);
  }
  HILC_CLEAR_ERROR();
  hipLaunchKernelGGL((resblock_kernel<C, STREAM, SCARRY>), dim3((unsigned)blocks), dim3(K::NT), 0, s, a);
  HILC_CHECK_LAUNCH();
  return HILC_OK;
}

}  // namespace
