// capi.hip -- the extern "C" boundary of include/city2ba_hip.h (gfx950 only).  ONE translation unit:
//   this file          includes, error plumbing, the kernel launchers (templates over the kernel instances), workspace entries
//   capi_level0.hpp    Level 0: stateless asynchronous launchers over device pointers; placed outputs; communicator
//   capi_host_rows.hpp host-side rows (CPU C++ behind the same ABI): layouts, candidates, mesh samplers, cull, files
//   capi_problem.hpp   Level 1: a BAProblem resident on one device (host buffers in / out, synchronous), *_sharded forms
#include "../../include/city2ba_hip.h"
#include "../../include/city2ba_hip_host.h"
#include "../../include/city2ba_hip_experimental.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "host_baproblem.hpp"
#include "host_bvh.hpp"
#include "host_generate.hpp"
#include "host_noise.hpp"
#include "host_synthetic.hpp"
#include "kernels.hpp"
#include "cull_kernels.hpp"
#include "cell_kernels.hpp"
#include "text_kernels.hpp"
#include "comm_rccl.hpp"

using namespace c2b;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                           \
    } while (0)

// A launch is checked by what it ADDS to the thread's error state: S() notes the error (if any) that was already
// pending before the launch -- left there by torch, RCCL or anyone else sharing the HIP runtime -- and LAUNCH_CHECK
// reports only an error that differs from it, consuming that one alone.  A pending foreign error is neither cleared
// nor blamed on this library.
thread_local hipError_t g_pending = hipSuccess;
#define LAUNCH_CHECK()                                                                                    \
    do {                                                                                                  \
        const hipError_t e_ = hipPeekAtLastError();                                                       \
        if (e_ != hipSuccess && e_ != g_pending) {                                                        \
            (void)hipGetLastError();                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "kernel launch: %s (%s:%d)",   \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                       \
        }                                                                                                 \
    } while (0)

// No C++ exception may cross the extern "C" boundary (std::vector / std::string allocate in the host-side rows): every
// int-returning entry point runs inside this pair and maps what it catches to a status code.
#define C2B_API_BEGIN try {
#define C2B_API_END(who)                                                                                  \
    } catch (const std::bad_alloc &) {                                                                    \
        return fail(C2B_ERR_OOM, who ": out of host memory");                                             \
    } catch (const std::exception &e_) {                                                                  \
        return fail(C2B_ERR_INVALID_ARGUMENT, who ": %s", e_.what());                                     \
    } catch (...) {                                                                                       \
        return fail(C2B_ERR_INVALID_ARGUMENT, who ": unknown C++ exception");                             \
    }

// Every launch passes its stream through S() (see LAUNCH_CHECK).
inline hipStream_t S(void *s) {
    g_pending = hipPeekAtLastError();
    return reinterpret_cast<hipStream_t>(s);
}
// the same rule for code that launches several kernels and checks once: the error those launches added, if any
inline hipError_t launch_error() {
    const hipError_t e = hipPeekAtLastError();
    if (e == hipSuccess || e == g_pending) return hipSuccess;
    (void)hipGetLastError();
    return e;
}
inline unsigned blocks_for(int64_t n, int b = kBlock) { return (unsigned)((n + b - 1) / b); }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// workspace layout (doubles): [stats records: kRedBlocks*kStatRec] [result slot of the fused error sum: 16]
//                            [ticket block: arrival counters of the in-kernel folds + magic, kTicketWords u32]
//                            [workgroup partials of the ticket fold: one per workgroup of the launch]
// The counters sit in the workspace they count partials of: a workspace serves one launch at a time (the partials
// already demand that), so launches from different streams, threads or graphs can never meet on the same counters.
// c2b_workspace_init zeroes them and writes the magic once; every fold leaves them zero.
constexpr int64_t kWsStatsDoubles = (int64_t)kRedBlocks * kStatRec;
constexpr int64_t kWsFinal = kWsStatsDoubles;
constexpr int64_t kWsTicket = kWsFinal + 16;                               // 128-byte aligned when the workspace is
constexpr int64_t kWsBlockPart = kWsTicket + kTicketWords / 2;
static_assert((kWsTicket * 8) % 128 == 0 && kTicketWords % 2 == 0, "ticket lines must stay 128-byte aligned");
// workgroups never hold fewer than 4 tiles of 64 observations (WPB * OPL >= 4 in every instantiation)
inline int64_t block_part_slots(int64_t n_obs) {
    const int64_t one_shot = ((n_obs + 63) / 64 + 3) / 4 + 8;
    return one_shot > 4096 + 8 ? one_shot : 4096 + 8;        // 4096: the persistent grids of the tuning library
}
inline unsigned *ws_ticket(void *workspace) {
    return workspace ? reinterpret_cast<unsigned *>(reinterpret_cast<double *>(workspace) + kWsTicket) : nullptr;
}


// one-shot launch of k_observations<MODE, NK, OPL, WPB>: one workgroup per WPB * OPL tiles of 64 observations
// CSR: cam_idx = the camera of every 64th observation, row_ptr / n_cam = the lists' boundaries (kernels.hpp)
template <int MODE, int OPL, int WPB, int MINW = 1, bool CSR = false, bool NTS = false, int NTL = 0>
void launch_obs_v(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                  const double *uv_obs, int64_t n, double norm, double max_dist, double *uv_out, uint8_t *keep,
                  double *block_part, unsigned *ticket, double *out_sum, hipStream_t st,
                  const uint64_t *row_ptr = nullptr, int64_t n_cam = 0, int64_t obs_base = 0, uint64_t seed = 0) {
    static_assert(WPB * OPL >= 8, "two partials per workgroup must fit block_part_slots (one per 4 tiles)");
    const int tiles = (int)(((n + 63) / 64 + WPB * OPL - 1) / (WPB * OPL));
#define C2B_GO(NK)                                                                                                      \
    hipLaunchKernelGGL((k_observations<MODE, NK, OPL, WPB, MINW, CSR, NTS, NTL>), dim3((unsigned)tiles), dim3(WPB * 64), 0, st,    \
                       camblk, reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx,                               \
                       reinterpret_cast<const double2 *>(uv_obs), (int)n, tiles, norm, max_dist,                       \
                       reinterpret_cast<double2 *>(uv_out), keep, block_part, ticket, out_sum, row_ptr, (int)n_cam, obs_base, seed)
    if constexpr (MODE != MODE_ERROR) { C2B_GO(NORM_2); }
    else if (norm == 2.0) C2B_GO(NORM_2);
    else if (norm == 1.0) C2B_GO(NORM_1);
    else C2B_GO(NORM_ANY);
#undef C2B_GO
}

// project / error sum / visibility predicate over an observation list.  MODE_ERROR folds the sum into out_sum
// (device pointer) in the same launch; it needs the workspace for its workgroup partials.
template <int MODE>
int launch_obs(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
               const double *uv_obs, int64_t n, double norm, double max_dist, double *uv_out, uint8_t *keep,
               void *workspace, double *out_sum, hipStream_t st, const uint64_t *row_ptr = nullptr, int64_t n_cam = 0,
               int64_t obs_base = 0, uint64_t seed = 0) {
    double *block_part = workspace ? reinterpret_cast<double *>(workspace) + kWsBlockPart : nullptr;
    unsigned *ticket = ws_ticket(workspace);
    // Cache policy of the streams (A/B in profiles/r02i_ab_cache_policy.txt): results leave through non-temporal stores
    // and the observed uv (read once) comes in through non-temporal loads, so that the tables every observation
    // gathers from -- points, cameras -- keep the L2 / Infinity Cache; the 4-byte point index stays cached where the
    // camera footprint is small (project, error: tables + indices of the --blocks 128 problem fit the 256 MB cache and
    // the next call finds them there) and is non-temporal where it is not (visibility, Jacobian: 256 B per camera).
    // (the fused observation noise reads and rewrites uv exactly once: both directions bypass the caches)
    constexpr int kNTL = (MODE == MODE_ERROR || MODE == MODE_ERROR12 || MODE == MODE_NOISE_ERROR12) ? 2 : ((MODE == MODE_VISIBILITY || MODE == MODE_VISIBILITY_BITS) ? 1 : 0);
    // the fused noise pass holds the draw's table entries and the camera chunks in flight at once: left alone the allocator takes
    // 68 registers (7 waves per SIMD, one workgroup per CU less); told to leave room for 8 waves it fits 64 without scratch
    constexpr int kMinW = (MODE == MODE_NOISE_ERROR12 || MODE == MODE_VISIBILITY || MODE == MODE_VISIBILITY_BITS) ? 8 : 1;
    if (row_ptr) {          // the *_rows entry points: cam_idx = the tile records of c2b_rows_pack
#define C2B_ROWS_ARGS camblk, pts4, cam_idx, pt_idx, uv_obs, n, norm, max_dist, uv_out, keep, block_part, ticket, out_sum, st, row_ptr, n_cam, obs_base, seed
        launch_obs_v<MODE, 3, 8, kMinW, true, true, kNTL>(C2B_ROWS_ARGS);
#undef C2B_ROWS_ARGS
        return C2B_OK;
    }
#define C2B_ARGS camblk, pts4, cam_idx, pt_idx, uv_obs, n, norm, max_dist, uv_out, keep, block_part, ticket, out_sum, st
    launch_obs_v<MODE, 3, 8, kMinW, false, true, kNTL>(C2B_ARGS);    // shipped (308): three tiles of 64 per wave
#undef C2B_ARGS
    return C2B_OK;
}


// CSR: cam_idx = the tile records of c2b_rows_pack for this launch's first observation (= observation obs_base of
// the list row_ptr describes)
template <bool WITH_ERR, int WPB, int OPL, int MINW, bool OBUP = true, bool CSR = false, int NTL = 0, bool NTS = true>
void launch_jac_l(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                  const double *uv_obs, int64_t n_obs, double *r, double *Jc, double *Jp, double norm,
                  double *block_part, unsigned *ticket, double *out_sum, hipStream_t st,
                  const uint64_t *row_ptr = nullptr, int64_t n_cam = 0, int64_t obs_base = 0) {
    // Two tiles per wave sit at the edge of 128 registers: the blocked camera table's address arithmetic tipped the allocator to 134
    // (three waves per SIMD, one workgroup per CU).  With the error sum the instance is told to leave room for four waves and stays
    // at 128 without scratch (tests/test_isa_pins.py); WITHOUT the sum the cap would spill 20 bytes, so a launch that wants no sum
    // takes the one-tile shape of the same workgroup count per CU instead (0 ... 2 % behind in fast output sets, nothing to fold).
    if constexpr (!WITH_ERR && WPB == 8 && OPL == 2 && MINW == 1) {
        launch_jac_l<WITH_ERR, 16, 1, MINW, OBUP, CSR, NTL, NTS>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, block_part, ticket,
                                                                     out_sum, st, row_ptr, n_cam, obs_base);
    } else {
    const int btiles = (int)(((n_obs + 63) / 64 + WPB * OPL - 1) / (WPB * OPL));
    constexpr int kMinW = (MINW == 1 && OPL == 2) ? 4 : MINW;
#define C2B_GO(NK)                                                                                                      \
    hipLaunchKernelGGL((k_residual_jacobian_l<NK, WITH_ERR, WPB, NTS, OPL, kMinW, OBUP, CSR, NTL>), dim3((unsigned)btiles),        \
                       dim3(WPB * 64), 0, st, camblk, reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx,        \
                       reinterpret_cast<const double2 *>(uv_obs), (int)n_obs, btiles, norm,                            \
                       reinterpret_cast<double2 *>(r), Jc, Jp, block_part, ticket, out_sum, row_ptr, (int)n_cam, obs_base)
    if constexpr (!WITH_ERR) { C2B_GO(NORM_2); }
    else if (norm == 2.0) C2B_GO(NORM_2);
    else if (norm == 1.0) C2B_GO(NORM_1);
    else C2B_GO(NORM_ANY);
#undef C2B_GO
    }
}

// Which of the Jacobian launch's once-read streams bypass the caches (0 none, 2 the observed uv, 3 uv and point index).
// The kernel gathers from two tables (camera records, points) and streams 20 bytes of input per observation; its 208
// bytes of results per observation always leave non-temporally.  When everything fits the 256 MB Infinity Cache the
// next launch finds its inputs there, so everything stays cached.  When the tables fit but tables + streams do not,
// cached streams would wash the tables out: the uv stream (16 B per observation) then goes non-temporal, and the point
// index too once tables + indices alone exceed the cache.  When not even the tables fit, nothing is gained and cached
// loads are 2 % faster.  Measured at --blocks 32 ... 208 (profiles/r02j_ab_jacobian_stream_policy.txt): the rule picks
// the fastest of the three at every size but one (--blocks 140: 1.4 % behind).  n_pts <= 0 (unknown): cached.
static int jacobian_stream_policy(int64_t n_obs, int64_t n_cam, int64_t n_pts) {
    constexpr int64_t kInfinityCache = 256ll << 20;                      // MI355X
    if (n_pts <= 0 || n_cam <= 0) return 0;
    const int64_t tables = n_cam * (int64_t)(kCamBlk * sizeof(double)) + n_pts * 32, idx = n_obs * 4, uv = n_obs * 16;
    if (tables > kInfinityCache || tables + idx + uv <= kInfinityCache) return 0;
    return tables + idx > kInfinityCache ? 3 : 2;
}

// Tiles of 64 observations per wave.  Two (both tiles' loads issued up front: the second tile's index -> gather chain
// hides behind the first tile's stores) is the better shape once there are enough workgroups to keep every CU busy
// through the last round; below ~6 M observations (< 24 workgroups of 1 024 per CU) the finer grain of one tile per
// wave wins: 2.4 M observations -- a rank's share of the headline problem at 8 GPUs -- 97.9 -> 94.3 us, 1.2 M 47.4 ->
// 46.4, 4.9 M 166.2 -> 164.3, against 332.5 -> 339.8 at 9.6 M (profiles/r02k_ab_jacobian_tiles_per_wave.txt).
constexpr int64_t kJacOneTileBelow = 6000000;

// ... and by the store rate of the output set, when the caller knows it (c2b_residual_jacobian_rows_placed: the rate
// c2b_jacobian_outputs_alloc measured).  MI355X devices / allocations take this launch's streaming stores at ~7.0-7.1
// TB/s or at ~5.6-5.9 TB/s, and the better shape differs: into a 7.1 TB/s set two tiles per wave in
// 512-thread workgroups is the fastest (692 us; 256 threads x one tile: 717, +3.6 %); into a 5.7-5.8 TB/s set -- on a
// device that has nothing faster and on the slow allocations of a mixed one alike -- ONE tile per wave in 256-thread
// workgroups is (842 against 862 us on a slow-store device, 825 / 845 and 836 / 856 in the slow sets of two mixed ones:
// -2.3 % every time, 1.06-1.08 x the launch's algorithmic bytes at the set's own store rate instead of 1.08-1.105;
// profiles/r05_ab_slow_store.txt, r05a/r05c_ab_step_*).  Stores that drain slowly keep a wave's registers and LDS busy
// longer; the finer grain -- half the observations per wave, a quarter per workgroup -- lets the CU turn over sooner.
// Between the classes (sets at 6.2-6.5 TB/s on mixed devices; profiles/r05p_ab_step_three_classes.txt and the two runs
// before it): 1 024 threads x ONE tile is the best there -- -3.6 / -3.2 / -2.7 % against 512 x 2 at 6.21 / 6.45 / 6.54 TB/s
// where 256 x 1 gives -4.4 / -1.9 / -1.6 % -- and it is the one shape that is never far off: -1 ... -3.4 % in the slow sets
// (256 x 1: -2.3 ... -4.4 %), 0 ... +1.8 % in the fast ones (256 x 1: +3.4 ... +3.7 %).  So three shapes by the measured rate:
constexpr double kJacSlowStoreGBs = 6300.0;       // below: 256 threads x 1 tile
constexpr double kJacFastStoreGBs = 6850.0;       // below: 1 024 threads x 1 tile; at or above (or unknown): 512 threads x 2 tiles
struct JacShape { int wpb, opl; };                // waves per workgroup, tiles of 64 observations per wave
// Below ~6 M observations (one tile per wave) the 1 024-thread workgroup -- ONE workgroup of 16 waves per CU instead of two of 8,
// half the workgroups to dispatch and to fold -- is 1-3.6 % faster than the 512-thread one at a rank's eighth of the headline
// problem (2.4 M observations: 108.9 / 113.0 us into a 5.4 TB/s set, 100.3 / 101.9 into a 6.1 TB/s one; 4.9 M: 217.5 / 223.1),
// on two devices, r05 (output sets of this size never reach the 7 TB/s class).
static JacShape jacobian_shape(int64_t n_obs, double store_GBs) {
    if (n_obs < kJacOneTileBelow) return {16, 1};
    if (store_GBs > 0.0 && store_GBs < kJacSlowStoreGBs) return {4, 1};
    if (store_GBs > 0.0 && store_GBs < kJacFastStoreGBs) return {16, 1};
    return {8, 2};
}

// residual + Jacobian; WITH_ERR also folds sum |r|^norm into out_sum (device pointer) in the same launch
template <bool WITH_ERR>
int launch_jacobian(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                    const double *uv_obs, int64_t n_obs, double *r, double *Jc, double *Jp, double norm, void *workspace,
                    double *out_sum, hipStream_t st, const uint64_t *row_ptr = nullptr, int64_t n_cam = 0, int64_t obs_base = 0,
                    int64_t n_pts = 0, double store_GBs = 0.0) {
    double *block_part = workspace ? reinterpret_cast<double *>(workspace) + kWsBlockPart : nullptr;
    unsigned *ticket = ws_ticket(workspace);
    if (row_ptr) {          // the *_rows entry points
#define C2B_ROWS_ARGS camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, block_part, ticket, out_sum, st, row_ptr, n_cam, obs_base
        const int policy = jacobian_stream_policy(n_obs, n_cam, n_pts);
        const JacShape shape = jacobian_shape(n_obs, store_GBs);
        if (shape.wpb == 4) {                                   // a slow-store output set: 256 threads x one tile
            switch (policy) {
                case 3: launch_jac_l<WITH_ERR, 4, 1, 1, true, true, 3>(C2B_ROWS_ARGS); break;
                case 2: launch_jac_l<WITH_ERR, 4, 1, 1, true, true, 2>(C2B_ROWS_ARGS); break;
                default: launch_jac_l<WITH_ERR, 4, 1, 1, true, true, 0>(C2B_ROWS_ARGS); break;
            }
        } else if (shape.opl == 1) {                           // below ~6 M observations, or a set between the store classes: 1 024 threads x one tile
            switch (policy) {
                case 3: launch_jac_l<WITH_ERR, 16, 1, 1, true, true, 3>(C2B_ROWS_ARGS); break;
                case 2: launch_jac_l<WITH_ERR, 16, 1, 1, true, true, 2>(C2B_ROWS_ARGS); break;
                default: launch_jac_l<WITH_ERR, 16, 1, 1, true, true, 0>(C2B_ROWS_ARGS); break;
            }
        } else {
            switch (policy) {
                case 3: launch_jac_l<WITH_ERR, 8, 2, 1, true, true, 3>(C2B_ROWS_ARGS); break;
                case 2: launch_jac_l<WITH_ERR, 8, 2, 1, true, true, 2>(C2B_ROWS_ARGS); break;
                default: launch_jac_l<WITH_ERR, 8, 2, 1, true, true, 0>(C2B_ROWS_ARGS); break;
            }
        }
#undef C2B_ROWS_ARGS
        return C2B_OK;
    }
#define C2B_ARGS camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, block_part, ticket, out_sum, st
    if (n_obs < kJacOneTileBelow) launch_jac_l<WITH_ERR, 16, 1, 1>(C2B_ARGS);  // shipped: lean form, one tile per wave in 1 024-thread workgroups (the grid of the rows form: same sum bits) ...
    else launch_jac_l<WITH_ERR, 8, 2, 1>(C2B_ARGS);                        // ... or two (= variant 40), by size
#undef C2B_ARGS
    return C2B_OK;
}

inline int stats_grid(int64_t n, int block = kBlock, int cap = kStatGrid) {
    static_assert(kStatGrid <= kRedBlocks, "one record slot per workgroup");
    int grid = (int)((n + block - 1) / block);
    if (grid > cap) grid = cap;
    return grid < 1 ? 1 : grid;
}

// mean / std / min / max / extent / origin in ONE launch folded by its last workgroup (kernels.hpp: Chan triples)
template <typename Src>
int stats_impl(const Src &src, int64_t n, void *workspace, double *stats, hipStream_t st) {
    double *rec = reinterpret_cast<double *>(workspace);
    const ShardMap whole{src.n_cam, 0, src.n_cam, 0};
    hipLaunchKernelGGL((k_stats_pass1<Src, true>), dim3(stats_grid(n, kStatBlock)), dim3(kStatBlock), 0, st, src, n, (double)n, rec,
                       ws_ticket(workspace), whole, stats);
    LAUNCH_CHECK();
    return C2B_OK;
}

template <typename T>
int drift_impl(const char *who, T *cam15, int64_t n_cam, T *pts4, int64_t n_pts, const double *origin,
               const double *stats_norm, double strength, double angle_strength, double std, double dx, double dy,
               double dz, uint64_t seed, hipStream_t st, int64_t cam_base = 0) {
    if (n_cam < 0 || n_pts < 0 || !origin || (n_cam && !cam15) || (n_pts && !pts4))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: bad arguments", who);
    if (!(std >= 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: std must be >= 0 (rand's Normal::new panics)", who);
    const int64_t n = n_cam + n_pts;
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_add_drift<T>, dim3(blocks_for(n)), dim3(kBlock), 0, st, cam15, n_cam,
                       reinterpret_cast<typename V4<T>::type *>(pts4), n_pts, origin, strength, angle_strength, std, dx,
                       dy, dz, stats_norm, seed, cam_base);
    LAUNCH_CHECK();
    return C2B_OK;
}

template <typename T>
int noise_entities_impl(const char *who, T *cam15, int64_t n_cam, T *pts4, int64_t n_pts, const double *stats,
                        double translation_std, double rotation_std, double point_std, uint64_t seed, hipStream_t st,
                        int64_t cam_base = 0) {
    if (n_cam < 0 || n_pts < 0 || !stats || (n_cam && !cam15) || (n_pts && !pts4))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: bad arguments", who);
    if (!(translation_std >= 0.0) || !(rotation_std >= 0.0) || !(point_std >= 0.0))
        return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise: standard deviations must be >= 0");
    const int64_t n = n_cam + n_pts;
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_add_noise_entities<T>, dim3(blocks_for(n)), dim3(kBlock), 0, st, cam15, n_cam,
                       reinterpret_cast<typename V4<T>::type *>(pts4), n_pts, stats, translation_std, rotation_std,
                       point_std, seed, cam_base);
    LAUNCH_CHECK();
    return C2B_OK;
}

template <typename T>
int sin_impl(const char *who, T *cam15, int64_t n_cam, T *pts4, int64_t n_pts, const double *stats, double dx, double dy,
             double dz, double nx, double ny, double nz, double strength, double frequency, hipStream_t st) {
    if (n_cam < 0 || n_pts < 0 || !stats || (n_cam && !cam15) || (n_pts && !pts4))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: bad arguments", who);
    const int64_t n = n_cam + n_pts;
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_add_sin_noise<T>, dim3(blocks_for(n)), dim3(kBlock), 0, st, cam15, n_cam,
                       reinterpret_cast<typename V4<T>::type *>(pts4), n_pts, stats, dx, dy, dz, nx, ny, nz, strength,
                       frequency);
    LAUNCH_CHECK();
    return C2B_OK;
}

}  // namespace

extern "C" {

const char *c2b_version(void) { return "city2ba_hip 0.6.0 (gfx950)"; }
int c2b_abi_version(void) { return C2B_ABI_VERSION; }
const char *c2b_last_error(void) { return g_err; }

int c2b_device_count(int *count) {
    C2B_API_BEGIN
    if (!count) return fail(C2B_ERR_INVALID_ARGUMENT, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(C2B_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return C2B_OK;
    C2B_API_END("device_count")
}

// Zero the arrival counters of the workspace's in-kernel folds and write the magic that marks it initialised
// (one tiny launch on `stream`; capture-safe).  Once per workspace, before its first use.
int c2b_workspace_init(void *workspace, void *stream) {
    C2B_API_BEGIN
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 127))
        return fail(C2B_ERR_INVALID_ARGUMENT, "workspace_init: workspace must be a 128-byte aligned device pointer");
    hipLaunchKernelGGL(k_workspace_init, dim3(1), dim3(256), 0, S(stream), ws_ticket(workspace));
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("workspace_init")
}

// Diagnostic: the arrival counters of a workspace must all be zero whenever no launch that uses it is in flight (the
// last workgroup of every fold resets them).  Synchronises `stream`, then counts the non-zero counter words; -1 if
// the workspace carries no magic (c2b_workspace_init was never called on it).
int c2b_workspace_selfcheck(const void *workspace, void *stream, int64_t *nonzero_words) {
    C2B_API_BEGIN
    if (!nonzero_words || !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "workspace_selfcheck: NULL argument");
    *nonzero_words = 0;
    HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    std::vector<unsigned> host((size_t)kTicketWords);
    HIP_TRY(hipMemcpy(host.data(), reinterpret_cast<const double *>(workspace) + kWsTicket, host.size() * sizeof(unsigned),
                      hipMemcpyDeviceToHost));
    if (host[kTicketMagicAt] != kTicketMagic) { *nonzero_words = -1; return C2B_OK; }
    int64_t c = 0;
    for (unsigned i = 0; i < kTicketWords; ++i) c += i != kTicketMagicAt && host[i] != 0;
    *nonzero_words = c;
    return C2B_OK;
    C2B_API_END("workspace_selfcheck")
}

int64_t c2b_workspace_bytes(int64_t n_obs) {
    if (n_obs < 0) n_obs = 0;
    return (kWsBlockPart + block_part_slots(n_obs)) * (int64_t)sizeof(double);
}


#include "capi_level0.hpp"
#include "capi_host_rows.hpp"
#include "capi_problem.hpp"

}  // extern "C"
