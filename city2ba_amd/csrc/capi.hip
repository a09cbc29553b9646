// capi.hip -- the extern "C" boundary of include/city2ba_hip.h (gfx950 only).
//
// Level 0: stateless asynchronous launchers over device pointers.
// Level 1: a BAProblem resident on one device, host buffers in / out, synchronous.
#include "../../include/city2ba_hip.h"

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
#ifdef C2B_TUNE
#include "obs_pipeline.hpp"      // persistent pipelined variants: measured slower, tuning library only
#endif
#include "cull_kernels.hpp"
#include "cell_kernels.hpp"
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

#ifdef C2B_TUNE
// Tuning build only (libcity2ba_hip_tune.so, tools/tune_*.py): kernel variants, including timing-only ablations
// whose outputs are wrong by construction.  None of this exists in the product library.
int g_jac_variant = 0;        // 0 = the shipped kernel
int g_obs_variant = 308;

template <typename K>
int persistent_grid(K kernel, int block_threads, int64_t work_blocks) {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cached[dev]) {
        int occ = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block_threads, 0) != hipSuccess || occ < 1) occ = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached[dev] = occ * cus;
    }
    int64_t g = cached[dev];
    if (g > work_blocks) g = work_blocks;
    g = (g + 7) & ~(int64_t)7;
    if (g < 8) g = 8;
    if (g > 4096) g = 4096;
    return (int)g;
}

template <int MODE, int WPB>
void launch_obs_p(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                  const double *uv_obs, int64_t n, double norm, double max_dist, double *uv_out, uint8_t *keep,
                  double *block_part, unsigned *ticket, double *out_sum, hipStream_t st) {
    const int64_t work = ((n >> 6) + WPB - 1) / WPB + 1;
#define C2B_GO(NK)                                                                                                      \
    do {                                                                                                                \
        const int grid = persistent_grid(k_observations_p<MODE, NK, WPB>, WPB * 64, work);                              \
        hipLaunchKernelGGL((k_observations_p<MODE, NK, WPB>), dim3(grid), dim3(WPB * 64), 0, st, camblk,                \
                           reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx,                                   \
                           reinterpret_cast<const double2 *>(uv_obs), (int)n, norm, max_dist,                          \
                           reinterpret_cast<double2 *>(uv_out), keep, block_part, ticket, out_sum);                     \
    } while (0)
    if constexpr (MODE != MODE_ERROR) { C2B_GO(NORM_2); }
    else if (norm == 2.0) C2B_GO(NORM_2);
    else if (norm == 1.0) C2B_GO(NORM_1);
    else C2B_GO(NORM_ANY);
#undef C2B_GO
}

template <bool WITH_ERR, int WPB, int MINW>
void launch_jac_p(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                  const double *uv_obs, int64_t n_obs, double *r, double *Jc, double *Jp, double norm,
                  double *block_part, unsigned *ticket, double *out_sum, hipStream_t st) {
    const int64_t work = ((n_obs >> 6) + WPB - 1) / WPB + 1;
#define C2B_GO(NK)                                                                                                      \
    do {                                                                                                                \
        const int grid = persistent_grid(k_residual_jacobian_p<NK, WITH_ERR, WPB, MINW, true>, WPB * 64, work);         \
        hipLaunchKernelGGL((k_residual_jacobian_p<NK, WITH_ERR, WPB, MINW, true>), dim3(grid), dim3(WPB * 64), 0, st,  \
                           camblk, reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx,                           \
                           reinterpret_cast<const double2 *>(uv_obs), (int)n_obs, norm, reinterpret_cast<double2 *>(r), \
                           Jc, Jp, block_part, ticket, out_sum);                                                        \
    } while (0)
    if constexpr (!WITH_ERR) { C2B_GO(NORM_2); }
    else if (norm == 2.0) C2B_GO(NORM_2);
    else if (norm == 1.0) C2B_GO(NORM_1);
    else C2B_GO(NORM_ANY);
#undef C2B_GO
}
#endif

// one-shot launch of k_observations<MODE, NK, OPL, WPB>: one workgroup per WPB * OPL tiles of 64 observations
// CSR: cam_idx = the camera of every 64th observation, row_ptr / n_cam = the lists' boundaries (kernels.hpp)
template <int MODE, int OPL, int WPB, int MINW = 1, bool FAKECI = false, bool CSR = false, bool NTS = false, int NTL = 0>
void launch_obs_v(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                  const double *uv_obs, int64_t n, double norm, double max_dist, double *uv_out, uint8_t *keep,
                  double *block_part, unsigned *ticket, double *out_sum, hipStream_t st,
                  const uint64_t *row_ptr = nullptr, int64_t n_cam = 0, int64_t obs_base = 0, uint64_t seed = 0) {
    static_assert(WPB * OPL >= 8, "two partials per workgroup must fit block_part_slots (one per 4 tiles)");
    const int tiles = (int)(((n + 63) / 64 + WPB * OPL - 1) / (WPB * OPL));
#define C2B_GO(NK)                                                                                                      \
    hipLaunchKernelGGL((k_observations<MODE, NK, OPL, WPB, MINW, FAKECI, CSR, NTS, NTL>), dim3((unsigned)tiles), dim3(WPB * 64), 0, st,    \
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
    constexpr int kNTL = (MODE == MODE_ERROR || MODE == MODE_ERROR12 || MODE == MODE_NOISE_ERROR12) ? 2 : (MODE == MODE_VISIBILITY ? 1 : 0);
    if (row_ptr) {          // the *_rows entry points: cam_idx = the tile records of c2b_rows_pack
#define C2B_ROWS_ARGS camblk, pts4, cam_idx, pt_idx, uv_obs, n, norm, max_dist, uv_out, keep, block_part, ticket, out_sum, st, row_ptr, n_cam, obs_base, seed
#ifdef C2B_TUNE
        switch (g_obs_variant) {
            case 20308: launch_obs_v<MODE, 3, 8, 1, false, true, false, 0>(C2B_ROWS_ARGS); return C2B_OK;   // everything cached (r02h)
            case 21308: launch_obs_v<MODE, 3, 8, 1, false, true, true, 0>(C2B_ROWS_ARGS); return C2B_OK;    // non-temporal stores only
            case 22308: launch_obs_v<MODE, 3, 8, 1, false, true, true, 2>(C2B_ROWS_ARGS); return C2B_OK;    // ... + observed uv
            case 23308: launch_obs_v<MODE, 3, 8, 1, false, true, true, 3>(C2B_ROWS_ARGS); return C2B_OK;    // ... + point index
            case 30108: launch_obs_v<MODE, 1, 8, 1, false, true, true, kNTL>(C2B_ROWS_ARGS); return C2B_OK;   // shipped policy, one tile per wave
            case 30208: launch_obs_v<MODE, 2, 8, 1, false, true, true, kNTL>(C2B_ROWS_ARGS); return C2B_OK;   // two
            case 30408: launch_obs_v<MODE, 4, 8, 1, false, true, true, kNTL>(C2B_ROWS_ARGS); return C2B_OK;   // four
            case 30304: launch_obs_v<MODE, 3, 4, 1, false, true, true, kNTL>(C2B_ROWS_ARGS); return C2B_OK;   // three, four waves per workgroup
            default: break;
        }
#endif
        launch_obs_v<MODE, 3, 8, 1, false, true, true, kNTL>(C2B_ROWS_ARGS);
#undef C2B_ROWS_ARGS
        return C2B_OK;
    }
#define C2B_ARGS camblk, pts4, cam_idx, pt_idx, uv_obs, n, norm, max_dist, uv_out, keep, block_part, ticket, out_sum, st
#ifdef C2B_TUNE
    switch (g_obs_variant) {
        case 108: launch_obs_v<MODE, 1, 8>(C2B_ARGS); return C2B_OK;
        case 216: launch_obs_v<MODE, 2, 16>(C2B_ARGS); return C2B_OK;
        case 204: launch_obs_v<MODE, 2, 4>(C2B_ARGS); return C2B_OK;
        case 208: launch_obs_v<MODE, 2, 8>(C2B_ARGS); return C2B_OK;
        case 1308: launch_obs_v<MODE, 3, 8, 8>(C2B_ARGS); return C2B_OK;    // 3 tiles per wave, registers capped for 8 waves per SIMD
        case 9308: launch_obs_v<MODE, 3, 8, 1, true>(C2B_ARGS); return C2B_OK;   // ablation: camera index computed, not loaded (wrong outputs)
        case 408: launch_obs_v<MODE, 4, 8>(C2B_ARGS); return C2B_OK;
        case 20308: launch_obs_v<MODE, 3, 8>(C2B_ARGS); return C2B_OK;                              // everything cached (r02h)
        case 23308: launch_obs_v<MODE, 3, 8, 1, false, false, true, 3>(C2B_ARGS); return C2B_OK;   // every stream non-temporal
        case 2004: launch_obs_p<MODE, 4>(C2B_ARGS); return C2B_OK;       // persistent pipelined forms (obs_pipeline.hpp)
        case 2008: launch_obs_p<MODE, 8>(C2B_ARGS); return C2B_OK;
        case 2016: launch_obs_p<MODE, 16>(C2B_ARGS); return C2B_OK;
        default: break;
    }
#endif
    launch_obs_v<MODE, 3, 8, 1, false, false, true, kNTL>(C2B_ARGS);    // shipped (308): three tiles of 64 per wave
#undef C2B_ARGS
    return C2B_OK;
}

#ifdef C2B_TUNE
template <bool WITH_ERR, int WPB, int SPLIT, bool NT, int ABL, int OPL, bool LDSCAM>
void launch_jac_w(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                  const double *uv_obs, int64_t n_obs, double *r, double *Jc, double *Jp, double norm,
                  double *block_part, unsigned *ticket, double *out_sum, hipStream_t st) {
    const int64_t wave_tiles = (n_obs + 63) / 64;
    const int64_t btiles = (wave_tiles + WPB * OPL - 1) / (WPB * OPL);
#define C2B_GO(NK)                                                                                                      \
    hipLaunchKernelGGL((k_residual_jacobian_w<NK, WITH_ERR, WPB, SPLIT, NT, ABL, OPL, LDSCAM>), dim3((unsigned)btiles), \
                       dim3(WPB * 64), 0, st, camblk, reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx,        \
                       reinterpret_cast<const double2 *>(uv_obs), n_obs, btiles, norm, reinterpret_cast<double2 *>(r), \
                       Jc, Jp, block_part, ticket, out_sum)
    if constexpr (!WITH_ERR) { C2B_GO(NORM_2); }
    else if (norm == 2.0) C2B_GO(NORM_2);
    else if (norm == 1.0) C2B_GO(NORM_1);
    else C2B_GO(NORM_ANY);
#undef C2B_GO
}
#endif

// CSR: cam_idx = the tile records of c2b_rows_pack for this launch's first observation (= observation obs_base of
// the list row_ptr describes)
template <bool WITH_ERR, int WPB, int OPL, int MINW, int XK = 0, bool OBUP = true, bool CSR = false, int NTL = 0>
void launch_jac_l(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                  const double *uv_obs, int64_t n_obs, double *r, double *Jc, double *Jp, double norm,
                  double *block_part, unsigned *ticket, double *out_sum, hipStream_t st,
                  const uint64_t *row_ptr = nullptr, int64_t n_cam = 0, int64_t obs_base = 0) {
    const int btiles = (int)(((n_obs + 63) / 64 + WPB * OPL - 1) / (WPB * OPL));
#define C2B_GO(NK)                                                                                                      \
    hipLaunchKernelGGL((k_residual_jacobian_l<NK, WITH_ERR, WPB, true, OPL, MINW, XK, OBUP, CSR, NTL>), dim3((unsigned)btiles),        \
                       dim3(WPB * 64), 0, st, camblk, reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx,        \
                       reinterpret_cast<const double2 *>(uv_obs), (int)n_obs, btiles, norm,                            \
                       reinterpret_cast<double2 *>(r), Jc, Jp, block_part, ticket, out_sum, row_ptr, (int)n_cam, obs_base)
    if constexpr (!WITH_ERR) { C2B_GO(NORM_2); }
    else if (norm == 2.0) C2B_GO(NORM_2);
    else if (norm == 1.0) C2B_GO(NORM_1);
    else C2B_GO(NORM_ANY);
#undef C2B_GO
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

// residual + Jacobian; WITH_ERR also folds sum |r|^norm into out_sum (device pointer) in the same launch
template <bool WITH_ERR>
int launch_jacobian(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                    const double *uv_obs, int64_t n_obs, double *r, double *Jc, double *Jp, double norm, void *workspace,
                    double *out_sum, hipStream_t st, const uint64_t *row_ptr = nullptr, int64_t n_cam = 0, int64_t obs_base = 0,
                    int64_t n_pts = 0) {
    double *block_part = workspace ? reinterpret_cast<double *>(workspace) + kWsBlockPart : nullptr;
    unsigned *ticket = ws_ticket(workspace);
    if (row_ptr) {          // the *_rows entry points
#define C2B_ROWS_ARGS camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, block_part, ticket, out_sum, st, row_ptr, n_cam, obs_base
#ifdef C2B_TUNE
        switch (g_jac_variant) {
            case 51: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 0>(C2B_ROWS_ARGS); return C2B_OK;    // every load cached
            case 52: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 2>(C2B_ROWS_ARGS); return C2B_OK;    // non-temporal observed uv
            case 53: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 3>(C2B_ROWS_ARGS); return C2B_OK;    // ... and point index
            case 61:                                                                                      // one tile per wave whatever the size, streams by the size rule
                switch (jacobian_stream_policy(n_obs, n_cam, n_pts)) {
                    case 3: launch_jac_l<WITH_ERR, 8, 1, 1, 0, true, true, 3>(C2B_ROWS_ARGS); break;
                    case 2: launch_jac_l<WITH_ERR, 8, 1, 1, 0, true, true, 2>(C2B_ROWS_ARGS); break;
                    default: launch_jac_l<WITH_ERR, 8, 1, 1, 0, true, true, 0>(C2B_ROWS_ARGS); break;
                }
                return C2B_OK;
            case 64:                                                                                      // two tiles per wave whatever the size
                switch (jacobian_stream_policy(n_obs, n_cam, n_pts)) {
                    case 3: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 3>(C2B_ROWS_ARGS); break;
                    case 2: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 2>(C2B_ROWS_ARGS); break;
                    default: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 0>(C2B_ROWS_ARGS); break;
                }
                return C2B_OK;
            default: break;
        }
#endif
        const int policy = jacobian_stream_policy(n_obs, n_cam, n_pts);
        if (n_obs < kJacOneTileBelow) {
            switch (policy) {
                case 3: launch_jac_l<WITH_ERR, 8, 1, 1, 0, true, true, 3>(C2B_ROWS_ARGS); break;
                case 2: launch_jac_l<WITH_ERR, 8, 1, 1, 0, true, true, 2>(C2B_ROWS_ARGS); break;
                default: launch_jac_l<WITH_ERR, 8, 1, 1, 0, true, true, 0>(C2B_ROWS_ARGS); break;
            }
        } else {
            switch (policy) {
                case 3: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 3>(C2B_ROWS_ARGS); break;
                case 2: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 2>(C2B_ROWS_ARGS); break;
                default: launch_jac_l<WITH_ERR, 8, 2, 1, 0, true, true, 0>(C2B_ROWS_ARGS); break;
            }
        }
#undef C2B_ROWS_ARGS
        return C2B_OK;
    }
#define C2B_ARGS camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, block_part, ticket, out_sum, st
#ifdef C2B_TUNE
    switch (g_jac_variant) {
        case 14: launch_jac_w<WITH_ERR, 8, 2, true, 0, 2, true>(C2B_ARGS); return C2B_OK;      // rounds 1-2a: two code paths (LDS / FLAT fallback), 64-bit indices
        case 9: launch_jac_w<WITH_ERR, 8, 2, true, 0, 1, false>(C2B_ARGS); return C2B_OK;
        case 13: launch_jac_w<WITH_ERR, 8, 2, true, 0, 2, false>(C2B_ARGS); return C2B_OK;     // FLAT camera reads
        case 16: launch_jac_w<WITH_ERR, 8, 2, true, 0, 1, true>(C2B_ARGS); return C2B_OK;
        case 17: launch_jac_w<WITH_ERR, 8, 2, false, 0, 2, true>(C2B_ARGS); return C2B_OK;    // shipped structure, plain (not nt) stores
        case 18: launch_jac_w<WITH_ERR, 4, 2, true, 0, 2, true>(C2B_ARGS); return C2B_OK;     // 256-thread workgroups
        case 31: {                                                                                  // store pattern only, plain stores
            const int64_t wt = (n_obs + 63) / 64, bt = (wt + 7) / 8;
            hipLaunchKernelGGL((k_store_pattern<false, 8>), dim3((unsigned)bt), dim3(512), 0, st, n_obs, bt, reinterpret_cast<double2 *>(r), Jc, Jp);
            return C2B_OK;
        }
        case 20: launch_jac_w<WITH_ERR, 8, 2, true, 1, 2, false>(C2B_ARGS); return C2B_OK;     // no Jacobian stores
        case 21: launch_jac_w<WITH_ERR, 8, 2, true, 2, 2, false>(C2B_ARGS); return C2B_OK;     // no arithmetic
        case 40: launch_jac_l<WITH_ERR, 8, 2, 1>(C2B_ARGS); return C2B_OK;      // lean form, observed uv requested up front
        case 49: launch_jac_l<WITH_ERR, 8, 2, 1, 0, false>(C2B_ARGS); return C2B_OK;   // ... requested per tile
        case 50: launch_jac_l<WITH_ERR, 8, 2, 4, 0, true>(C2B_ARGS); return C2B_OK;    // up front, registers capped for 4 waves per SIMD
        case 42: launch_jac_l<WITH_ERR, 8, 1, 1>(C2B_ARGS); return C2B_OK;      // one tile per wave
        case 44: launch_jac_l<WITH_ERR, 8, 2, 1, 1>(C2B_ARGS); return C2B_OK;     // tiles in launch order (no XCD-aware map)
        case 45: launch_jac_l<WITH_ERR, 8, 2, 1, 4>(C2B_ARGS); return C2B_OK;     // chunked XCD map, K = 4
        case 46: launch_jac_l<WITH_ERR, 8, 2, 1, 16>(C2B_ARGS); return C2B_OK;    // K = 16
        case 47: launch_jac_l<WITH_ERR, 8, 2, 1, 64>(C2B_ARGS); return C2B_OK;    // K = 64
        case 48: launch_jac_l<WITH_ERR, 8, 2, 1, 256>(C2B_ARGS); return C2B_OK;   // K = 256
        case 32: {                                                                 // store pattern only, tiles in launch order
            const int64_t wt = (n_obs + 63) / 64, bt = (wt + 7) / 8;
            hipLaunchKernelGGL((k_store_pattern<true, 8, false>), dim3((unsigned)bt), dim3(512), 0, st, n_obs, bt, reinterpret_cast<double2 *>(r), Jc, Jp);
            return C2B_OK;
        }
        case 43: launch_jac_l<WITH_ERR, 8, 3, 1>(C2B_ARGS); return C2B_OK;      // three tiles per wave
        case 100: launch_jac_p<WITH_ERR, 8, 4>(C2B_ARGS); return C2B_OK;     // persistent pipelined forms (obs_pipeline.hpp)
        case 104: launch_jac_p<WITH_ERR, 4, 1>(C2B_ARGS); return C2B_OK;     // 12 waves per CU at the natural register count
        case 105: launch_jac_p<WITH_ERR, 4, 4>(C2B_ARGS); return C2B_OK;     // 16 waves per CU (spills)
        case 108: launch_jac_p<WITH_ERR, 8, 1>(C2B_ARGS); return C2B_OK;     //  8 waves per CU
        case 116: launch_jac_p<WITH_ERR, 16, 4>(C2B_ARGS); return C2B_OK;    // 16 waves per CU, one workgroup
        default: break;
    }
#endif
    if (n_obs < kJacOneTileBelow) launch_jac_l<WITH_ERR, 8, 1, 1>(C2B_ARGS);   // shipped: lean form, one tile per wave (= variant 42) ...
    else launch_jac_l<WITH_ERR, 8, 2, 1>(C2B_ARGS);                        // ... or two (= variant 40), by size
#undef C2B_ARGS
    return C2B_OK;
}

inline int stats_grid(int64_t n) {
    int grid = (int)((n + kBlock - 1) / kBlock);
    if (grid > kRedBlocks) grid = kRedBlocks;
    return grid < 1 ? 1 : grid;
}

// mean / std / min / max / extent / origin in ONE launch folded by its last workgroup (kernels.hpp: Chan triples)
template <typename Src>
int stats_impl(const Src &src, int64_t n, void *workspace, double *stats, hipStream_t st) {
    double *rec = reinterpret_cast<double *>(workspace);
    const ShardMap whole{src.n_cam, 0, src.n_cam, 0};
    hipLaunchKernelGGL((k_stats_pass1<Src, true>), dim3(stats_grid(n)), dim3(kBlock), 0, st, src, n, (double)n, rec,
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

const char *c2b_version(void) { return "city2ba_hip 0.1.0 (gfx950)"; }
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

#ifdef C2B_TUNE
// tuning hooks of libcity2ba_hip_tune.so (tools/tune_jac.py, tools/tune_obs.py); absent from the product library
int c2b_tune_set_jacobian_variant(int v) { g_jac_variant = v; return C2B_OK; }
int c2b_tune_set_observation_variant(int v) { g_obs_variant = v; return C2B_OK; }
#endif

/* ------------------------------- level 0 --------------------------------------------- */

int c2b_cameras_from_bal(const double *bal9, int64_t n, double *cam15, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!bal9 || !cam15))) return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_from_bal: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_cameras_from_bal, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), bal9, n, cam15);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_from_bal")
}

int c2b_cameras_to_bal(const double *cam15, int64_t n, double *bal9, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!bal9 || !cam15))) return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_to_bal: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_cameras_to_bal, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, n, bal9);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_to_bal")
}

int c2b_cameras_prepare_state(const double *cam15, int64_t n, double *camblk, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!cam15 || !camblk))) return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_prepare_state: bad arguments");
    if (!n) return C2B_OK;
    if (!aligned16(camblk)) return fail(C2B_ERR_INVALID_ARGUMENT, "camblk must be 16-byte aligned");
    hipLaunchKernelGGL(k_cameras_prepare<false>, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, n, camblk);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_prepare_state")
}

int c2b_cameras_prepare_bal(const double *bal9, int64_t n, double *camblk, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!bal9 || !camblk))) return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_prepare_bal: bad arguments");
    if (!n) return C2B_OK;
    if (!aligned16(camblk)) return fail(C2B_ERR_INVALID_ARGUMENT, "camblk must be 16-byte aligned");
    hipLaunchKernelGGL(k_cameras_prepare<true>, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), bal9, n, camblk);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_prepare_bal")
}

int c2b_cameras_from_position_direction(const double *pos3, const double *dir9, int64_t n, double *cam15,
                                        void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!pos3 || !dir9 || !cam15)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_from_position_direction: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_cameras_from_position_direction, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), pos3, dir9,
                       n, cam15);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_from_position_direction")
}

int c2b_project_world(const double *cam15, const uint32_t *cam_idx, const double *p3, int64_t n, double *out3,
                      void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!cam15 || !cam_idx || !p3 || !out3))) return fail(C2B_ERR_INVALID_ARGUMENT, "project_world: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_camera_point_map<false>, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, cam_idx, p3, n, out3);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("project_world")
}

int c2b_to_world(const double *cam15, const uint32_t *cam_idx, const double *p3, int64_t n, double *out3, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!cam15 || !cam_idx || !p3 || !out3))) return fail(C2B_ERR_INVALID_ARGUMENT, "to_world: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_camera_point_map<true>, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, cam_idx, p3, n, out3);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("to_world")
}

int c2b_cameras_transform(double *cam15, const double *delta_dir9, const double *delta_loc3, int64_t n, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!cam15 || !delta_dir9 || !delta_loc3))) return fail(C2B_ERR_INVALID_ARGUMENT, "cameras_transform: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_cameras_transform, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), cam15, delta_dir9, delta_loc3, n);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("cameras_transform")
}

int c2b_points_pad(const double *pts3, int64_t n, double *pts4, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!pts3 || !pts4))) return fail(C2B_ERR_INVALID_ARGUMENT, "points_pad: bad arguments");
    if (!n) return C2B_OK;
    if (!aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "pts4 must be 16-byte aligned");
    hipLaunchKernelGGL(k_points_pad, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), pts3, n,
                       reinterpret_cast<double4 *>(pts4));
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("points_pad")
}

int c2b_points_unpad(const double *pts4, int64_t n, double *pts3, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!pts3 || !pts4))) return fail(C2B_ERR_INVALID_ARGUMENT, "points_unpad: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_points_unpad, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream),
                       reinterpret_cast<const double4 *>(pts4), n, pts3);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("points_unpad")
}

int c2b_expand_rows(const uint64_t *row_ptr, int64_t n_cam, int64_t obs_base, int64_t n_obs,
                    uint32_t *cam_idx, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_obs < 0 || obs_base < 0 || (n_obs && (!row_ptr || !cam_idx)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "expand_rows: bad arguments");
    if (n_cam >= (int64_t)1 << 32) return fail(C2B_ERR_INVALID_ARGUMENT, "expand_rows: n_cam exceeds u32");
    if (!n_obs) return C2B_OK;
    hipLaunchKernelGGL(k_expand_rows, dim3(blocks_for(n_obs)), dim3(kBlock), 0, S(stream), row_ptr, n_cam,
                       obs_base, n_obs, cam_idx);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("expand_rows")
}

static int check_obs_args(const char *who, const void *camblk, const void *pts4, const void *cam_idx,
                          const void *pt_idx, int64_t n) {
    if (n < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: negative count", who);
    if (n > (int64_t)0x7fffffff - 4096 * 64)       // 32-bit observation indices on the device; 2^31 observations are 34 GB of indices and uv alone
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: more than 2^31 observations in one launch", who);
    if (n && (!camblk || !pts4 || !cam_idx || !pt_idx)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: NULL input", who);
    if (n && (!aligned16(camblk) || !aligned16(pts4)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: camblk/pts4 must be 16-byte aligned", who);
    return C2B_OK;
}

int c2b_project(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                int64_t n_obs, double *uv_out, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("project", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!uv_out || !aligned16(uv_out)) return fail(C2B_ERR_INVALID_ARGUMENT, "project: uv_out NULL or misaligned");
    rc = launch_obs<MODE_PROJECT>(camblk, pts4, cam_idx, pt_idx, nullptr, n_obs, 0.0, 0.0, uv_out, nullptr, nullptr, nullptr, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("project")
}

int c2b_reprojection_error_sum(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                               const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs, double norm,
                               void *workspace, double *out_sum, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("reprojection_error_sum", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!out_sum) return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sum: out_sum is NULL");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream))); return C2B_OK; }
    if (!uv_obs || !aligned16(uv_obs) || !workspace)
        return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sum: uv_obs/workspace NULL or misaligned");
    rc = launch_obs<MODE_ERROR>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, norm, 0.0, nullptr, nullptr, workspace, out_sum, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("reprojection_error_sum")
}

// ---- camera-major lists addressed through the row structure (no per-observation camera index) ----
int64_t c2b_rows_tiles_bytes(int64_t n_obs) { return n_obs <= 0 ? 0 : (n_obs + 63) / 64 * 16; }

static int check_rows_args(const char *who, const uint64_t *row_ptr, int64_t n_cam, const void *tiles, int64_t n) {
    if (n_cam < 0 || n_cam >= (int64_t)1 << 31) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: n_cam out of range", who);
    if (n && (!row_ptr || !tiles || n_cam == 0)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: NULL row_ptr / tiles, or no cameras", who);
    if (n && (!aligned16(tiles) || (reinterpret_cast<uintptr_t>(row_ptr) & 7)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s: tiles must be 16-byte aligned, row_ptr 8-byte aligned", who);
    return C2B_OK;
}

int c2b_rows_pack(const uint64_t *row_ptr, int64_t n_cam, int64_t n_obs, void *tiles, void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || n_obs > (int64_t)0x7fffffff - 4096 * 64) return fail(C2B_ERR_INVALID_ARGUMENT, "rows_pack: observation count out of range");
    int rc = check_rows_args("rows_pack", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    const int64_t n_tiles = (n_obs + 63) / 64;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(k_rows_pack_tiles, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, st, row_ptr, (int)n_cam, (int)n_obs,
                       reinterpret_cast<uint4 *>(tiles));
    hipLaunchKernelGGL(k_rows_pack_marks, dim3((unsigned)((n_cam + 255) / 256)), dim3(256), 0, st, row_ptr, (int)n_cam, (int)n_obs,
                       reinterpret_cast<uint32_t *>(tiles));
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("rows_pack")
}

int c2b_project_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam, const void *tiles,
                     const uint32_t *pt_idx, int64_t n_obs, double *uv_out, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("project_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("project_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!uv_out || !aligned16(uv_out)) return fail(C2B_ERR_INVALID_ARGUMENT, "project_rows: uv_out NULL or misaligned");
    rc = launch_obs<MODE_PROJECT>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, nullptr, n_obs, 0.0, 0.0, uv_out,
                                  nullptr, nullptr, nullptr, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("project_rows")
}

int c2b_reprojection_error_sum_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                                    const void *tiles, const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                                    double norm, void *workspace, double *out_sum, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("reprojection_error_sum_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("reprojection_error_sum_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!out_sum) return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sum_rows: out_sum is NULL");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream))); return C2B_OK; }
    if (!uv_obs || !aligned16(uv_obs) || !workspace)
        return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sum_rows: uv_obs/workspace NULL or misaligned");
    rc = launch_obs<MODE_ERROR>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, uv_obs, n_obs, norm, 0.0, nullptr,
                                nullptr, workspace, out_sum, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("reprojection_error_sum_rows")
}

// L1 and L2 in one pass: out_sums[0] = sum |du| + |dv|, out_sums[1] = sum du^2 + dv^2 -- each bit-identical to what
// c2b_reprojection_error_sum_rows returns for that norm (same grid, same fold order per sum, one arrival count).
int c2b_reprojection_error_sums2_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                                      const void *tiles, const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs,
                                      void *workspace, double *out_sums, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("reprojection_error_sums2_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("reprojection_error_sums2_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!out_sums) return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sums2_rows: out_sums is NULL");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sums, 0, 2 * sizeof(double), S(stream))); return C2B_OK; }
    if (!uv_obs || !aligned16(uv_obs) || !workspace)
        return fail(C2B_ERR_INVALID_ARGUMENT, "reprojection_error_sums2_rows: uv_obs/workspace NULL or misaligned");
    rc = launch_obs<MODE_ERROR12>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, uv_obs, n_obs, 0.0, 0.0, nullptr,
                                  nullptr, workspace, out_sums, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("reprojection_error_sums2_rows")
}

// add_noise's observation pass (src/noise.rs:152-170) and the two error sums run_noise evaluates right after it
// (src/bin/city2ba.rs:350-354) in ONE pass over the list: uv is perturbed in place exactly as
// c2b_add_noise_observations would (same draws: counter = obs_base + i), and out_sums = the L1 / L2 sums of the
// PERTURBED observations against the cameras and points as they are now (entity noise first, then this).
int c2b_add_noise_observations_error_sums2_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam,
                                                const void *tiles, const uint32_t *pt_idx, double *uv, int64_t n_obs,
                                                int64_t obs_base, double observations_std, uint64_t seed, void *workspace,
                                                double *out_sums, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("add_noise_observations_error_sums2_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("add_noise_observations_error_sums2_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (!out_sums || obs_base < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_observations_error_sums2_rows: bad arguments");
    if (!(observations_std >= 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise: standard deviations must be >= 0");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sums, 0, 2 * sizeof(double), S(stream))); return C2B_OK; }
    if (!uv || !aligned16(uv) || !workspace)
        return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_observations_error_sums2_rows: uv/workspace NULL or misaligned");
    rc = launch_obs<MODE_NOISE_ERROR12>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, nullptr, n_obs, observations_std,
                                        0.0, uv, nullptr, workspace, out_sums, S(stream), row_ptr, n_cam, obs_base, seed);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("add_noise_observations_error_sums2_rows")
}

int c2b_visibility_rows(const double *camblk, const double *pts4, const uint64_t *row_ptr, int64_t n_cam, const void *tiles,
                        const uint32_t *pt_idx, int64_t n_pairs, double max_dist, double *uv_out, uint8_t *keep, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("visibility_rows", camblk, pts4, tiles, pt_idx, n_pairs);
    if (!rc) rc = check_rows_args("visibility_rows", row_ptr, n_cam, tiles, n_pairs);
    if (rc) return rc;
    if (!n_pairs) return C2B_OK;
    if (!uv_out || !keep || !aligned16(uv_out)) return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_rows: NULL or misaligned output");
    rc = launch_obs<MODE_VISIBILITY>(camblk, pts4, reinterpret_cast<const uint32_t *>(tiles), pt_idx, nullptr, n_pairs, 0.0, max_dist,
                                     uv_out, keep, nullptr, nullptr, S(stream), row_ptr, n_cam);
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_rows")
}

int c2b_residual_jacobian(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                          const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs, double *r, double *Jc,
                          double *Jp, double norm, void *workspace, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("residual_jacobian", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!uv_obs || !r || !Jc || !Jp) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian: NULL buffer");
    if (!aligned16(uv_obs) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian: uv/r/Jc/Jp must be 16-byte aligned");
    // with a workspace the fused error sum lands in the workspace's result slot (c2b_error_sum_finish copies it out)
    double *slot = workspace ? reinterpret_cast<double *>(workspace) + kWsFinal : nullptr;
    if (workspace) rc = launch_jacobian<true>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, workspace, slot, S(stream));
    else rc = launch_jacobian<false>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, nullptr, nullptr, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("residual_jacobian")
}

int c2b_residual_jacobian_sum(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                              const uint32_t *pt_idx, const double *uv_obs, int64_t n_obs, double *r, double *Jc,
                              double *Jp, double norm, void *workspace, double *out_sum, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("residual_jacobian_sum", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!out_sum) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_sum: out_sum is NULL");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream))); return C2B_OK; }
    if (!uv_obs || !r || !Jc || !Jp || !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_sum: NULL buffer");
    if (!aligned16(uv_obs) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_sum: uv/r/Jc/Jp must be 16-byte aligned");
    rc = launch_jacobian<true>(camblk, pts4, cam_idx, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, workspace, out_sum, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("residual_jacobian_sum")
}

int c2b_jacobian_stream_policy(int64_t n_obs, int64_t n_cam, int64_t n_pts) { return jacobian_stream_policy(n_obs, n_cam, n_pts); }
int c2b_jacobian_tiles_per_wave(int64_t n_obs) { return n_obs < kJacOneTileBelow ? 1 : 2; }

int c2b_residual_jacobian_rows(const double *camblk, const double *pts4, int64_t n_pts, const uint64_t *row_ptr, int64_t n_cam,
                               const void *tiles, int64_t obs_base, const uint32_t *pt_idx, const double *uv_obs,
                               int64_t n_obs, double *r, double *Jc, double *Jp, double norm, void *workspace,
                               double *out_sum, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("residual_jacobian_rows", camblk, pts4, tiles, pt_idx, n_obs);
    if (!rc) rc = check_rows_args("residual_jacobian_rows", row_ptr, n_cam, tiles, n_obs);
    if (rc) return rc;
    if (obs_base < 0 || (obs_base & 63)) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows: obs_base must be a non-negative multiple of 64");
    if (out_sum && !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows: out_sum needs a workspace");
    if (!n_obs) {
        if (out_sum) HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream)));
        return C2B_OK;
    }
    if (!uv_obs || !r || !Jc || !Jp) return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows: NULL buffer");
    if (!aligned16(uv_obs) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "residual_jacobian_rows: uv/r/Jc/Jp must be 16-byte aligned");
    const uint32_t *rec = reinterpret_cast<const uint32_t *>(tiles);
    if (workspace) {
        double *dst = out_sum ? out_sum : reinterpret_cast<double *>(workspace) + kWsFinal;
        rc = launch_jacobian<true>(camblk, pts4, rec, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, workspace, dst, S(stream), row_ptr, n_cam, obs_base, n_pts);
    } else {
        rc = launch_jacobian<false>(camblk, pts4, rec, pt_idx, uv_obs, n_obs, r, Jc, Jp, norm, nullptr, nullptr, S(stream), row_ptr, n_cam, obs_base, n_pts);
    }
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("residual_jacobian_rows")
}

int c2b_calib_store_pattern(int64_t n_obs, double *r, double *Jc, double *Jp, void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || (n_obs && (!r || !Jc || !Jp)) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "calib_store_pattern: bad arguments");
    if (n_obs < 64) return C2B_OK;
    const int64_t wt = (n_obs + 63) / 64, bt = (wt + 7) / 8;
    hipLaunchKernelGGL((k_store_pattern<true, 8>), dim3((unsigned)bt), dim3(512), 0, S(stream), n_obs, bt,
                       reinterpret_cast<double2 *>(r), Jc, Jp);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("calib_store_pattern")
}

int c2b_calib_store_pattern_map(int64_t n_obs, double *r, double *Jc, double *Jp, int tile_map, void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || tile_map < 0 || (n_obs && (!r || !Jc || !Jp)) || !aligned16(r) || !aligned16(Jc) || !aligned16(Jp))
        return fail(C2B_ERR_INVALID_ARGUMENT, "calib_store_pattern_map: bad arguments");
    if (n_obs < 64) return C2B_OK;
    const int64_t wt = (n_obs + 63) / 64, bt = (wt + 7) / 8;
    hipLaunchKernelGGL((k_store_pattern_map<true, 8>), dim3((unsigned)bt), dim3(512), 0, S(stream), n_obs, bt, tile_map,
                       reinterpret_cast<double2 *>(r), Jc, Jp);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("calib_store_pattern_map")
}

int c2b_calib_copy(const void *src, void *dst, int64_t bytes, void *stream) {
    C2B_API_BEGIN
    if (bytes < 0 || (bytes && (!src || !dst)) || !aligned16(src) || !aligned16(dst) || (bytes & 15))
        return fail(C2B_ERR_INVALID_ARGUMENT, "calib_copy: NULL, misaligned or not a multiple of 16 bytes");
    if (!bytes) return C2B_OK;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, S(stream), reinterpret_cast<const double2 *>(src),
                       reinterpret_cast<double2 *>(dst), bytes / 16);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("calib_copy")
}

// ---- output arrays of the residual + Jacobian launch, placed for streaming stores -------------------------------
// Measured on MI355X ("what the spread really is"): the same kernel writing the same bytes takes
// 690 or 860 us depending only on WHICH device allocation r / Jc / Jp live in -- the store pattern alone streams at
// ~7.0 TB/s into some allocations and ~5.7 TB/s into others of identical size and alignment, in one process on one
// device; a freed and re-made allocation keeps its speed, a different one rolls again.  Nothing visible from user
// space predicts it, so the placement is chosen by measurement: allocate, time the kernel's own store pattern
// (~0.6 ms per repetition), keep the set if it streams at fast_store_GBs or better, otherwise HOLD it (so that the
// allocator cannot hand the same memory back) and try again; the best of max_attempts wins, the rest are freed.
// Held memory is bounded (max_attempts <= 8 sets of 208 B per observation) and an out-of-memory attempt ends the
// search with the best set so far instead of failing.
struct c2b_jacobian_outputs {
    int device = 0;
    int64_t n_obs = 0;
    double *r = nullptr, *Jc = nullptr, *Jp = nullptr;
    int attempts = 0, chosen = -1;
    double rate[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {
struct OutSet {
    double *r = nullptr, *Jc = nullptr, *Jp = nullptr;
    void free_all() { if (r) (void)hipFree(r); if (Jc) (void)hipFree(Jc); if (Jp) (void)hipFree(Jp); r = Jc = Jp = nullptr; }
};
hipError_t alloc_set(int64_t n, OutSet *s) {
    const size_t k = (size_t)(n > 0 ? n : 1);
    hipError_t e = hipMalloc((void **)&s->r, k * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&s->Jc, k * 144);
    if (e == hipSuccess) e = hipMalloc((void **)&s->Jp, k * 48);
    if (e != hipSuccess) s->free_all();
    return e;
}
}  // namespace

int c2b_jacobian_outputs_alloc(int64_t n_obs, int max_attempts, double fast_store_GBs, void *stream, c2b_jacobian_outputs **out) {
    C2B_API_BEGIN
    if (!out || n_obs < 0 || n_obs >= ((int64_t)1 << 31)) return fail(C2B_ERR_INVALID_ARGUMENT, "jacobian_outputs_alloc: bad arguments");
    *out = nullptr;
    if (max_attempts < 1) max_attempts = 1;
    if (max_attempts > 8) max_attempts = 8;
    if (!(fast_store_GBs > 0.0)) fast_store_GBs = 7000.0;
    hipStream_t st = S(stream);
    std::unique_ptr<c2b_jacobian_outputs> h(new c2b_jacobian_outputs);
    HIP_TRY(hipGetDevice(&h->device));
    h->n_obs = n_obs;
    OutSet sets[8];
    auto free_sets = [&]() { for (auto &q : sets) q.free_all(); };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool measure = n_obs >= 1000000 && max_attempts > 1;      // below that the store rate means nothing
    if (measure && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) {
        if (e0) (void)hipEventDestroy(e0);
        return fail(C2B_ERR_HIP, "jacobian_outputs_alloc: hipEventCreate failed");
    }
    int best = -1;
    hipError_t err = hipSuccess;
    for (int a = 0; a < (measure ? max_attempts : 1); ++a) {
        err = alloc_set(n_obs, &sets[a]);
        if (err != hipSuccess) {
            if (best >= 0 && err == hipErrorOutOfMemory) { (void)hipGetLastError(); err = hipSuccess; }   // keep the best so far
            break;
        }
        h->attempts = a + 1;
        if (!measure) { best = a; break; }
        const int64_t wt = (n_obs + 63) / 64, bt = (wt + 7) / 8;
        auto pattern = [&]() {
            hipLaunchKernelGGL((k_store_pattern<true, 8>), dim3((unsigned)bt), dim3(512), 0, st, n_obs, bt,
                               reinterpret_cast<double2 *>(sets[a].r), sets[a].Jc, sets[a].Jp);
        };
        pattern(); pattern();
        err = hipEventRecord(e0, st);
        for (int k = 0; k < 4; ++k) pattern();
        if (err == hipSuccess) err = hipEventRecord(e1, st);
        if (err == hipSuccess) err = hipEventSynchronize(e1);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        if (err == hipSuccess) err = launch_error();
        if (err != hipSuccess) break;
        h->rate[a] = (double)n_obs * 208.0 / ((double)ms / 4.0 * 1e-3) / 1e9;
        // a later set replaces the incumbent only if it is clearly faster (2 %): between sets of the same class the
        // measured rate differs by noise, and the kernel's own time does not follow differences that small
        if (best < 0 || h->rate[a] > h->rate[best] * 1.02) best = a;
        if (h->rate[a] >= fast_store_GBs) break;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (err != hipSuccess || best < 0) {
        free_sets();
        return fail(err == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "jacobian_outputs_alloc: %s",
                    hipGetErrorString(err == hipSuccess ? hipErrorUnknown : err));
    }
    h->r = sets[best].r; h->Jc = sets[best].Jc; h->Jp = sets[best].Jp;
    sets[best] = OutSet();
    free_sets();
    h->chosen = best;
    *out = h.release();
    return C2B_OK;
    C2B_API_END("jacobian_outputs_alloc")
}

int c2b_jacobian_outputs_pointers(const c2b_jacobian_outputs *h, double **r, double **Jc, double **Jp) {
    if (!h || !r || !Jc || !Jp) return fail(C2B_ERR_INVALID_ARGUMENT, "jacobian_outputs_pointers: NULL argument");
    *r = h->r; *Jc = h->Jc; *Jp = h->Jp;
    return C2B_OK;
}

int c2b_jacobian_outputs_log(const c2b_jacobian_outputs *h, double *store_GBs_per_attempt, int capacity, int *attempts, int *chosen) {
    if (!h || capacity < 0 || (capacity && !store_GBs_per_attempt)) return fail(C2B_ERR_INVALID_ARGUMENT, "jacobian_outputs_log: bad arguments");
    for (int a = 0; a < h->attempts && a < capacity; ++a) store_GBs_per_attempt[a] = h->rate[a];
    if (attempts) *attempts = h->attempts;
    if (chosen) *chosen = h->chosen;
    return C2B_OK;
}

void c2b_jacobian_outputs_free(c2b_jacobian_outputs *h) {
    if (!h) return;
    int prev = 0;
    const bool sw = hipGetDevice(&prev) == hipSuccess && prev != h->device && hipSetDevice(h->device) == hipSuccess;
    if (h->r) (void)hipFree(h->r);
    if (h->Jc) (void)hipFree(h->Jc);
    if (h->Jp) (void)hipFree(h->Jp);
    if (sw) (void)hipSetDevice(prev);
    delete h;
}

int c2b_error_sum_finish(const void *workspace, int64_t n_obs, double *out_sum, void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || !out_sum) return fail(C2B_ERR_INVALID_ARGUMENT, "error_sum_finish: bad arguments");
    if (!n_obs) { HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(double), S(stream))); return C2B_OK; }
    if (!workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "error_sum_finish: workspace is NULL");
    HIP_TRY(hipMemcpyAsync(out_sum, reinterpret_cast<const double *>(workspace) + kWsFinal, sizeof(double),
                           hipMemcpyDeviceToDevice, S(stream)));
    return C2B_OK;
    C2B_API_END("error_sum_finish")
}

int c2b_visibility_pairs(const double *camblk, const double *pts4, const uint32_t *cam_idx,
                         const uint32_t *pt_idx, int64_t n_pairs, double max_dist, double *uv_out,
                         uint8_t *keep, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("visibility_pairs", camblk, pts4, cam_idx, pt_idx, n_pairs);
    if (rc) return rc;
    if (!n_pairs) return C2B_OK;
    if (!uv_out || !keep || !aligned16(uv_out)) return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_pairs: NULL/misaligned output");
    rc = launch_obs<MODE_VISIBILITY>(camblk, pts4, cam_idx, pt_idx, nullptr, n_pairs, 0.0, max_dist, uv_out, keep, nullptr, nullptr, S(stream));
    if (rc) return rc;
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_pairs")
}

int c2b_occlusion_filter(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                         int64_t n_obs, const float *tri9, int64_t n_tri, uint8_t *keep, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("occlusion_filter", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!keep || n_tri < 0 || (n_tri && !tri9)) return fail(C2B_ERR_INVALID_ARGUMENT, "occlusion_filter: bad arguments");
    hipLaunchKernelGGL(k_occlusion, dim3(blocks_for(n_obs)), dim3(kBlock), 0, S(stream), camblk,
                       reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx, n_obs, tri9, n_tri, keep);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("occlusion_filter")
}

struct c2b_bvh {
    c2b_host::Bvh b;
};

int c2b_bvh_build(const float *tri9, int64_t n_tri, c2b_bvh **out) {
    C2B_API_BEGIN
    if (!out || n_tri < 0 || (n_tri && !tri9)) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_build: bad arguments");
    *out = nullptr;
    if (n_tri >= ((int64_t)1 << 28)) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_build: more than 2^28 triangles");
    for (int64_t k = 0; k < 9 * n_tri; ++k)
        if (!std::isfinite(tri9[k])) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_build: triangle %lld is not finite", (long long)(k / 9));
    c2b_bvh *h = new (std::nothrow) c2b_bvh();
    if (!h) return fail(C2B_ERR_OOM, "bvh_build: host allocation failed");
    try {
        c2b_host::bvh_build(tri9, n_tri, h->b);
    } catch (const std::bad_alloc &) {
        delete h;
        return fail(C2B_ERR_OOM, "bvh_build: out of host memory");
    }
    if (h->b.depth >= kBvhStack) {
        const int d = h->b.depth;
        delete h;
        return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_build: hierarchy depth %d exceeds the traversal stack", d);
    }
    *out = h;
    return C2B_OK;
    C2B_API_END("bvh_build")
}

int c2b_bvh_sizes(const c2b_bvh *b, int64_t *n_nodes, int64_t *n_slots, int *depth) {
    C2B_API_BEGIN
    if (!b) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_sizes: bvh is NULL");
    if (n_nodes) *n_nodes = (int64_t)b->b.nodes.size();
    if (n_slots) *n_slots = (int64_t)b->b.order.size();
    if (depth) *depth = b->b.depth;
    return C2B_OK;
    C2B_API_END("bvh_sizes")
}

int c2b_bvh_copy(const c2b_bvh *b, void *nodes, void *tris, uint32_t *order) {
    C2B_API_BEGIN
    if (!b) return fail(C2B_ERR_INVALID_ARGUMENT, "bvh_copy: bvh is NULL");
    if (nodes) std::memcpy(nodes, b->b.nodes.data(), b->b.nodes.size() * sizeof(c2b_host::BvhNode));
    if (tris && !b->b.tris.empty()) std::memcpy(tris, b->b.tris.data(), b->b.tris.size() * sizeof(float));
    if (order && !b->b.order.empty()) std::memcpy(order, b->b.order.data(), b->b.order.size() * sizeof(uint32_t));
    return C2B_OK;
    C2B_API_END("bvh_copy")
}

void c2b_bvh_free(c2b_bvh *b) { delete b; }

int c2b_occlusion_filter_bvh(const double *camblk, const double *pts4, const uint32_t *cam_idx, const uint32_t *pt_idx,
                             int64_t n_obs, const void *nodes, int64_t n_nodes, const void *tris, int64_t n_slots,
                             uint8_t *keep, uint32_t *overflow, void *stream) {
    C2B_API_BEGIN
    int rc = check_obs_args("occlusion_filter_bvh", camblk, pts4, cam_idx, pt_idx, n_obs);
    if (rc) return rc;
    if (!n_obs) return C2B_OK;
    if (!keep || !overflow || !nodes || n_nodes < 1 || n_slots < 0 || (n_slots && !tris) || !aligned16(nodes) || (tris && !aligned16(tris)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "occlusion_filter_bvh: NULL/misaligned buffer");
    hipLaunchKernelGGL(k_occlusion_bvh, dim3(blocks_for(n_obs)), dim3(kBlock), 0, S(stream), camblk,
                       reinterpret_cast<const double4 *>(pts4), cam_idx, pt_idx, n_obs, reinterpret_cast<const float4 *>(nodes),
                       reinterpret_cast<const float4 *>(tris), keep, overflow);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("occlusion_filter_bvh")
}

int c2b_stats(const double *camblk, int64_t n_cam, const double *pts4, int64_t n_pts, void *workspace,
              double *stats, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || !stats || !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: bad arguments");
    if (n_cam + n_pts == 0) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: empty problem (the reference's fold1().unwrap() panics)");
    if ((n_cam && !camblk) || (n_pts && !pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: NULL input");
    if (!aligned16(camblk) || !aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: camblk/pts4 must be 16-byte aligned");
    const SrcBlk src{camblk, reinterpret_cast<const double4 *>(pts4), n_cam};
    return stats_impl(src, n_cam + n_pts, workspace, stats, S(stream));
    C2B_API_END("stats")
}

int c2b_stats_partial_pass1(const double *camblk, int64_t n_cam, int64_t cam_base, int64_t n_cam_global,
                            const double *pts4, int64_t n_pts, int64_t pt_base, int64_t n_entities_global,
                            void *workspace, double *part, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || cam_base < 0 || pt_base < 0 || n_cam_global < cam_base + n_cam || n_entities_global < 1 ||
        !part || !workspace || (n_cam && !camblk) || (n_pts && !pts4))
        return fail(C2B_ERR_INVALID_ARGUMENT, "stats_partial_pass1: bad arguments");
    if (!aligned16(camblk) || !aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_partial_pass1: camblk/pts4 must be 16-byte aligned");
    const SrcBlk src{camblk, reinterpret_cast<const double4 *>(pts4), n_cam};
    const int64_t n = n_cam + n_pts;
    double *rec = reinterpret_cast<double *>(workspace);
    const ShardMap map{n_cam, cam_base, n_cam_global, pt_base};
    hipLaunchKernelGGL((k_stats_pass1<SrcBlk, false>), dim3(stats_grid(n)), dim3(kBlock), 0, S(stream), src, n, (double)n_entities_global, rec,
                       ws_ticket(workspace), map, part);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("stats_partial_pass1")
}

int c2b_stats_partial_pass2(const double *camblk, int64_t n_cam, const double *pts4, int64_t n_pts, const double *mean3,
                            void *workspace, double *sumsq3, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || !mean3 || !sumsq3 || !workspace || (n_cam && !camblk) || (n_pts && !pts4))
        return fail(C2B_ERR_INVALID_ARGUMENT, "stats_partial_pass2: bad arguments");
    if (!aligned16(camblk) || !aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_partial_pass2: camblk/pts4 must be 16-byte aligned");
    const SrcBlk src{camblk, reinterpret_cast<const double4 *>(pts4), n_cam};
    const int64_t n = n_cam + n_pts;
    double *rec = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL((k_stats_pass2<SrcBlk, true>), dim3(stats_grid(n)), dim3(kBlock), 0, S(stream), src, n, mean3, rec,
                       ws_ticket(workspace), sumsq3);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("stats_partial_pass2")
}

// ---- collectives of the sharded path (comm_rccl.hpp): RCCL behind the C ABI -----------------------------------
#define RCCL_TRY(who, expr)                                                                                  \
    do {                                                                                                     \
        const ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                               \
            return fail(C2B_ERR_RCCL, who ": %s: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
    } while (0)
#define NEED_RCCL(who)                                                                                       \
    if (!rccl().ok()) return fail(C2B_ERR_RCCL, who ": %s", rccl().error.c_str())

const char *c2b_comm_backend(void) {
    static thread_local char text[256];
    if (!rccl().ok()) { snprintf(text, sizeof text, "unavailable: %s", rccl().error.c_str()); return text; }
    int v = 0;
    (void)rccl().GetVersion(&v);
    snprintf(text, sizeof text, "RCCL %d.%d.%d (%s)", v / 10000, (v / 100) % 100, v % 100, rccl().path.c_str());
    return text;
}

int c2b_comm_unique_id(void *id128) {
    C2B_API_BEGIN
    if (!id128) return fail(C2B_ERR_INVALID_ARGUMENT, "comm_unique_id: NULL argument");
    NEED_RCCL("comm_unique_id");
    static_assert(sizeof(ncclUniqueId) == C2B_COMM_ID_BYTES, "C2B_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id;
    RCCL_TRY("comm_unique_id", rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return C2B_OK;
    C2B_API_END("comm_unique_id")
}

int c2b_comm_init_rank(const void *id128, int rank, int world, int device, c2b_comm **out) {
    C2B_API_BEGIN
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world || device < 0)
        return fail(C2B_ERR_INVALID_ARGUMENT, "comm_init_rank: bad arguments");
    *out = nullptr;
    NEED_RCCL("comm_init_rank");
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    std::unique_ptr<c2b_comm> c(new c2b_comm);
    c->rank = rank; c->world = world; c->device = device;
    RCCL_TRY("comm_init_rank", rccl().CommInitRank(&c->comm, world, id, rank));
    *out = c.release();
    return C2B_OK;
    C2B_API_END("comm_init_rank")
}

int c2b_comm_init_all(int n_dev, const int *dev_ids, c2b_comm **out) {
    C2B_API_BEGIN
    if (n_dev < 1 || n_dev > 64 || !out) return fail(C2B_ERR_INVALID_ARGUMENT, "comm_init_all: bad arguments");
    for (int i = 0; i < n_dev; ++i) out[i] = nullptr;
    NEED_RCCL("comm_init_all");
    int devs[64];
    ncclComm_t comms[64];
    for (int i = 0; i < n_dev; ++i) devs[i] = dev_ids ? dev_ids[i] : i;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    const ncclResult_t r = rccl().CommInitAll(comms, n_dev, devs);      // switches the current device as it goes
    (void)hipSetDevice(prev);
    if (r != ncclSuccess) return fail(C2B_ERR_RCCL, "comm_init_all: ncclCommInitAll: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
    int made = 0;
    for (; made < n_dev; ++made) {
        out[made] = new (std::nothrow) c2b_comm;
        if (!out[made]) break;
        out[made]->comm = comms[made]; out[made]->rank = made; out[made]->world = n_dev; out[made]->device = devs[made];
    }
    if (made < n_dev) {                                       // out of host memory half way: give every communicator back
        for (int i = 0; i < made; ++i) { c2b_comm_destroy(out[i]); out[i] = nullptr; }
        for (int i = made; i < n_dev; ++i) {
            const bool sw = hipSetDevice(devs[i]) == hipSuccess;
            (void)rccl().CommDestroy(comms[i]);
            if (sw) (void)hipSetDevice(prev);
        }
        (void)hipSetDevice(prev);
        return fail(C2B_ERR_OOM, "comm_init_all: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("comm_init_all")
}

int c2b_comm_info(const c2b_comm *c, int *rank, int *world, int *device) {
    if (!c) return fail(C2B_ERR_INVALID_ARGUMENT, "comm_info: NULL communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return C2B_OK;
}

int c2b_comm_group_start(void) {
    C2B_API_BEGIN
    NEED_RCCL("comm_group_start");
    RCCL_TRY("comm_group_start", rccl().GroupStart());
    return C2B_OK;
    C2B_API_END("comm_group_start")
}

int c2b_comm_group_end(void) {
    C2B_API_BEGIN
    NEED_RCCL("comm_group_end");
    RCCL_TRY("comm_group_end", rccl().GroupEnd());
    return C2B_OK;
    C2B_API_END("comm_group_end")
}

int c2b_comm_all_reduce_sum_f64(c2b_comm *c, double *buf, int64_t n, void *stream) {
    C2B_API_BEGIN
    if (!c || !c->comm || n < 0 || (n && !buf)) return fail(C2B_ERR_INVALID_ARGUMENT, "comm_all_reduce_sum_f64: bad arguments");
    if (!n) return C2B_OK;
    RCCL_TRY("comm_all_reduce_sum_f64", rccl().AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, c->comm, S(stream)));
    return C2B_OK;
    C2B_API_END("comm_all_reduce_sum_f64")
}

int c2b_comm_all_gather_f64(c2b_comm *c, const double *send, int64_t n_per_rank, double *recv, void *stream) {
    C2B_API_BEGIN
    if (!c || !c->comm || n_per_rank < 0 || (n_per_rank && (!send || !recv)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "comm_all_gather_f64: bad arguments");
    if (!n_per_rank) return C2B_OK;
    RCCL_TRY("comm_all_gather_f64", rccl().AllGather(send, recv, (size_t)n_per_rank, ncclDouble, c->comm, S(stream)));
    return C2B_OK;
    C2B_API_END("comm_all_gather_f64")
}

void c2b_comm_destroy(c2b_comm *c) {
    if (!c) return;
    if (c->comm && rccl().ok()) {
        int prev = 0;
        const bool sw = hipGetDevice(&prev) == hipSuccess && prev != c->device && hipSetDevice(c->device) == hipSuccess;
        (void)rccl().CommDestroy(c->comm);
        if (sw) (void)hipSetDevice(prev);
    }
    delete c;
}

// ---- host halves of the statistics over sharded cameras ------------------------------------
// shares [world][20] = every rank's c2b_stats_partial_pass1 record in rank order.  mean: the shares summed in rank
// order; origin: smallest distance, ties to the LARGER global index (fold1 with strict <, src/noise.rs:80-86).
int c2b_stats_combine_shares(const double *shares, int world, double *stats) {
    C2B_API_BEGIN
    if (!shares || !stats || world < 1) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_combine_shares: bad arguments");
    const double inf = std::numeric_limits<double>::infinity();
    double mean[3] = {0, 0, 0}, mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    const double *best = nullptr;
    for (int r = 0; r < world; ++r) {
        const double *p = shares + 20 * (size_t)r;
        for (int k = 0; k < 3; ++k) {
            mean[k] = mean[k] + p[k];
            mn[k] = std::fmin(mn[k], p[6 + k]);
            mx[k] = std::fmax(mx[k], p[9 + k]);
        }
        if (p[18] < 0) continue;
        if (!best || p[19] < best[19] || (p[19] == best[19] && p[18] > best[18])) best = p;
    }
    if (!best) return fail(C2B_ERR_INVALID_ARGUMENT, "stats: empty problem");
    for (int k = 0; k < 20; ++k) stats[k] = 0.0;
    for (int k = 0; k < 3; ++k) {
        stats[k] = mean[k]; stats[6 + k] = mn[k]; stats[9 + k] = mx[k]; stats[12 + k] = mx[k] - mn[k];
        stats[15 + k] = best[15 + k];
    }
    stats[18] = best[18];
    stats[19] = best[19];
    return C2B_OK;
    C2B_API_END("stats_combine_shares")
}

// sumsq [world][3] = every rank's c2b_stats_partial_pass2 sums, rank order -> stats[3..5] = std, stats[19] = |std|
int c2b_stats_finish_shares(const double *sumsq, int world, int64_t n_entities, double *stats) {
    C2B_API_BEGIN
    if (!sumsq || !stats || world < 1 || n_entities < 1) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_finish_shares: bad arguments");
    double t[3] = {0, 0, 0};
    for (int r = 0; r < world; ++r)
        for (int k = 0; k < 3; ++k) t[k] = t[k] + sumsq[3 * (size_t)r + k];
    const double num = (double)n_entities;
    for (int k = 0; k < 3; ++k) stats[3 + k] = std::sqrt(t[k] / num);
    stats[19] = std::sqrt((stats[3] * stats[3] + stats[4] * stats[4]) + stats[5] * stats[5]);
    return C2B_OK;
    C2B_API_END("stats_finish_shares")
}

// BAProblem::mean/std/extent/dimensions + add_drift's origin when cameras are sharded: this rank's camblk holds cameras
// [cam_base, cam_base + n_cam) of n_cam_global, pts4 is the whole replicated table and rank r of W reduces its r-th
// slice.  Two all-gathers (20 and 3 doubles per rank) through the communicator; sums in rank order on the host, so
// every rank ends with the same bits.  Synchronous; `stats` (device, 20 doubles) is complete on return.
int c2b_stats_sharded(c2b_comm *c, const double *camblk, int64_t n_cam, int64_t cam_base, int64_t n_cam_global,
                      const double *pts4, int64_t n_pts, void *workspace, double *stats, void *stream) {
    C2B_API_BEGIN
    if (!c || !c->comm || !workspace || !stats || n_cam < 0 || n_pts < 0 || cam_base < 0 || n_cam_global < cam_base + n_cam)
        return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: bad arguments");
    if (!aligned16(camblk) || !aligned16(pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: camblk/pts4 must be 16-byte aligned");
    const int W = c->world, R = c->rank;
    const int64_t lo = n_pts * R / W, hi = n_pts * (R + 1) / W, n_ent = n_cam_global + n_pts;
    if (n_ent < 1) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: empty problem");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device)
        return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: the current device (%d) is not the communicator's (%d)", cur, c->device);
    hipStream_t st = S(stream);
    // scratch [20 mine | W x 20 | 3 mean | 3 mine | W x 3] carved out of the workspace's partial slots, which the
    // statistics kernels do not use (they keep their records in front of them): no device allocation per call -- a
    // hipMalloc / hipFree pair synchronises the device, ~1 ms each, while the peers' collectives are in flight
    const size_t n_dev = 20 + 20 * (size_t)W + 3 + 3 + 3 * (size_t)W;
    if ((int64_t)n_dev > block_part_slots(0)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_sharded: more than %d ranks", 150);
    double *dev = reinterpret_cast<double *>(workspace) + kWsBlockPart;
    double *d_mine = dev, *d_all = dev + 20, *d_mean = d_all + 20 * (size_t)W, *d_sq = d_mean + 3, *d_sqall = d_sq + 3;
    std::vector<double> shares(20 * (size_t)W), sq(3 * (size_t)W);
    double host_stats[20];
    for (double &v : host_stats) v = std::numeric_limits<double>::quiet_NaN();
    // A rank that fails locally keeps taking part in BOTH all-gathers (its peers are already waiting in them) and
    // reports its first error afterwards: `first` carries it.
    int first = C2B_OK;
    char first_msg[sizeof g_err] = "";
    auto note = [&](int rc) { if (rc && !first) { first = rc; std::snprintf(first_msg, sizeof first_msg, "%s", g_err); } return rc; };
    auto hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess) note(fail(C2B_ERR_HIP, "stats_sharded: %s: %s", what, hipGetErrorString(e)));
        return e == hipSuccess;
    };
    note(c2b_stats_partial_pass1(camblk, n_cam, cam_base, n_cam_global, pts4 + 4 * lo, hi - lo, lo, n_ent, workspace, d_mine, stream));
    note(c2b_comm_all_gather_f64(c, d_mine, 20, d_all, stream));
    if (hip(hipMemcpyAsync(shares.data(), d_all, shares.size() * sizeof(double), hipMemcpyDeviceToHost, st), "copy of the shares") &&
        hip(hipStreamSynchronize(st), "synchronize") && !first)
        note(c2b_stats_combine_shares(shares.data(), W, host_stats));
    hip(hipMemcpyAsync(d_mean, host_stats, 3 * sizeof(double), hipMemcpyHostToDevice, st), "upload of the mean");
    if (!first) note(c2b_stats_partial_pass2(camblk, n_cam, pts4 + 4 * lo, hi - lo, d_mean, workspace, d_sq, stream));
    note(c2b_comm_all_gather_f64(c, d_sq, 3, d_sqall, stream));
    if (hip(hipMemcpyAsync(sq.data(), d_sqall, sq.size() * sizeof(double), hipMemcpyDeviceToHost, st), "copy of the squared sums") &&
        hip(hipStreamSynchronize(st), "synchronize") && !first)
        note(c2b_stats_finish_shares(sq.data(), W, n_ent, host_stats));
    if (first) { std::snprintf(g_err, sizeof g_err, "%s", first_msg); return first; }
    HIP_TRY(hipMemcpyAsync(stats, host_stats, sizeof host_stats, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return C2B_OK;
    C2B_API_END("stats_sharded")
}

int c2b_add_drift_sharded(double *cam15, int64_t n_cam, int64_t cam_base, double *pts4, int64_t n_pts, const double *stats,
                          int normalized, double strength, double angle_strength, double std, double dir_x, double dir_y,
                          double dir_z, uint64_t seed, void *stream) {
    C2B_API_BEGIN
    if (!stats || cam_base < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "add_drift_sharded: bad arguments");
    return drift_impl<double>("add_drift_sharded", cam15, n_cam, pts4, n_pts, stats + 15, normalized ? stats : nullptr,
                              strength, angle_strength, std, dir_x, dir_y, dir_z, seed, S(stream), cam_base);
    C2B_API_END("add_drift_sharded")
}

int c2b_add_noise_entities_sharded(double *cam15, int64_t n_cam, int64_t cam_base, double *pts4, int64_t n_pts,
                                   const double *stats, double translation_std, double rotation_std, double point_std,
                                   uint64_t seed, void *stream) {
    C2B_API_BEGIN
    if (cam_base < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_entities_sharded: bad arguments");
    return noise_entities_impl<double>("add_noise_entities_sharded", cam15, n_cam, pts4, n_pts, stats, translation_std,
                                       rotation_std, point_std, seed, S(stream), cam_base);
    C2B_API_END("add_noise_entities_sharded")
}

int c2b_stats_f32(const float *cam15, int64_t n_cam, const float *pts4, int64_t n_pts, void *workspace,
                  double *stats, void *stream) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || !stats || !workspace) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_f32: bad arguments");
    if (n_cam + n_pts == 0) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_f32: empty problem");
    if ((n_cam && !cam15) || (n_pts && !pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "stats_f32: NULL input");
    const SrcState32 src{cam15, reinterpret_cast<const float4 *>(pts4), n_cam};
    return stats_impl(src, n_cam + n_pts, workspace, stats, S(stream));
    C2B_API_END("stats_f32")
}

int c2b_convert_f64_to_f32(const double *src, int64_t n, float *dst, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!src || !dst))) return fail(C2B_ERR_INVALID_ARGUMENT, "convert_f64_to_f32: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_f64_to_f32, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), src, n, dst);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("convert_f64_to_f32")
}

int c2b_convert_f32_to_f64(const float *src, int64_t n, double *dst, void *stream) {
    C2B_API_BEGIN
    if (n < 0 || (n && (!src || !dst))) return fail(C2B_ERR_INVALID_ARGUMENT, "convert_f32_to_f64: bad arguments");
    if (!n) return C2B_OK;
    hipLaunchKernelGGL(k_f32_to_f64, dim3(blocks_for(n)), dim3(kBlock), 0, S(stream), src, n, dst);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("convert_f32_to_f64")
}

int c2b_add_drift(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *origin,
                  double strength, double angle_strength, double std, double dir_x, double dir_y, double dir_z,
                  uint64_t seed, void *stream) {
    C2B_API_BEGIN
    return drift_impl<double>("add_drift", cam15, n_cam, pts4, n_pts, origin, nullptr, strength, angle_strength, std,
                              dir_x, dir_y, dir_z, seed, S(stream));
    C2B_API_END("add_drift")
}
int c2b_add_drift_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *origin,
                      double strength, double angle_strength, double std, double dir_x, double dir_y, double dir_z,
                      uint64_t seed, void *stream) {
    C2B_API_BEGIN
    return drift_impl<float>("add_drift_f32", cam15, n_cam, pts4, n_pts, origin, nullptr, strength, angle_strength, std,
                             dir_x, dir_y, dir_z, seed, S(stream));
    C2B_API_END("add_drift_f32")
}
int c2b_add_drift_normalized(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *stats,
                             double strength, double angle_strength, double std, uint64_t seed, void *stream) {
    C2B_API_BEGIN
    if (!stats) return fail(C2B_ERR_INVALID_ARGUMENT, "add_drift_normalized: stats is NULL");
    return drift_impl<double>("add_drift_normalized", cam15, n_cam, pts4, n_pts, stats + 15, stats, strength,
                              angle_strength, std, 0.0, 0.0, 0.0, seed, S(stream));
    C2B_API_END("add_drift_normalized")
}
int c2b_add_drift_normalized_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *stats,
                                 double strength, double angle_strength, double std, uint64_t seed, void *stream) {
    C2B_API_BEGIN
    if (!stats) return fail(C2B_ERR_INVALID_ARGUMENT, "add_drift_normalized_f32: stats is NULL");
    return drift_impl<float>("add_drift_normalized_f32", cam15, n_cam, pts4, n_pts, stats + 15, stats, strength,
                             angle_strength, std, 0.0, 0.0, 0.0, seed, S(stream));
    C2B_API_END("add_drift_normalized_f32")
}

int c2b_add_noise_entities(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *stats,
                           double translation_std, double rotation_std, double point_std, uint64_t seed,
                           void *stream) {
    C2B_API_BEGIN
    return noise_entities_impl<double>("add_noise_entities", cam15, n_cam, pts4, n_pts, stats, translation_std,
                                       rotation_std, point_std, seed, S(stream));
    C2B_API_END("add_noise_entities")
}
int c2b_add_noise_entities_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *stats,
                               double translation_std, double rotation_std, double point_std, uint64_t seed,
                               void *stream) {
    C2B_API_BEGIN
    return noise_entities_impl<float>("add_noise_entities_f32", cam15, n_cam, pts4, n_pts, stats, translation_std,
                                      rotation_std, point_std, seed, S(stream));
    C2B_API_END("add_noise_entities_f32")
}

int c2b_add_noise_observations(double *uv, int64_t n_obs, int64_t obs_base, double observations_std, uint64_t seed,
                               void *stream) {
    C2B_API_BEGIN
    if (n_obs < 0 || obs_base < 0 || (n_obs && !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_observations: bad arguments");
    if (!(observations_std >= 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise: standard deviations must be >= 0");
    if (!n_obs) return C2B_OK;
    if (!aligned16(uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_noise_observations: uv must be 16-byte aligned");
    hipLaunchKernelGGL(k_add_noise_observations, dim3(blocks_for(n_obs)), dim3(kBlock), 0, S(stream),
                       reinterpret_cast<double2 *>(uv), n_obs, obs_base, observations_std, seed);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("add_noise_observations")
}

int c2b_add_sin_noise(double *cam15, int64_t n_cam, double *pts4, int64_t n_pts, const double *stats, double dir_x,
                      double dir_y, double dir_z, double ndir_x, double ndir_y, double ndir_z, double strength,
                      double frequency, void *stream) {
    C2B_API_BEGIN
    return sin_impl<double>("add_sin_noise", cam15, n_cam, pts4, n_pts, stats, dir_x, dir_y, dir_z, ndir_x, ndir_y, ndir_z,
                            strength, frequency, S(stream));
    C2B_API_END("add_sin_noise")
}
int c2b_add_sin_noise_f32(float *cam15, int64_t n_cam, float *pts4, int64_t n_pts, const double *stats, double dir_x,
                          double dir_y, double dir_z, double ndir_x, double ndir_y, double ndir_z, double strength,
                          double frequency, void *stream) {
    C2B_API_BEGIN
    return sin_impl<float>("add_sin_noise_f32", cam15, n_cam, pts4, n_pts, stats, dir_x, dir_y, dir_z, ndir_x, ndir_y,
                           ndir_z, strength, frequency, S(stream));
    C2B_API_END("add_sin_noise_f32")
}

int c2b_partition_cameras(const uint64_t *row_ptr, int64_t n_cam, int n_parts, int64_t *bounds) {
    C2B_API_BEGIN
    if (!row_ptr || !bounds || n_cam < 0 || n_parts < 1) return fail(C2B_ERR_INVALID_ARGUMENT, "partition_cameras: bad arguments");
    const uint64_t total = row_ptr[n_cam];
    bounds[0] = 0;
    int64_t c = 0;
    for (int k = 1; k < n_parts; ++k) {
        // first camera whose prefix reaches k/n_parts of the observations
        const uint64_t target = (uint64_t)(((__uint128_t)total * (unsigned)k) / (unsigned)n_parts);
        int64_t lo = c, hi = n_cam;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (row_ptr[mid] < target) lo = mid + 1; else hi = mid;
        }
        c = lo;
        bounds[k] = c;
    }
    bounds[n_parts] = n_cam;
    return C2B_OK;
    C2B_API_END("partition_cameras")
}

int64_t c2b_visibility_dense_tiles(int64_t n_pts) { return n_pts <= 0 ? 0 : (n_pts + kDenseTile - 1) / kDenseTile; }

static void dense_grid(int64_t n_cam, int64_t n_tiles, dim3 *grid, int64_t *cams_per_chunk) {
    const int64_t bx = (n_tiles + kDenseWPB - 1) / kDenseWPB;
    // enough waves to fill 256 CUs a few times over, camera chunks in multiples of the LDS tile
    int64_t chunks = (16384 + n_tiles - 1) / (n_tiles > 0 ? n_tiles : 1);
    const int64_t max_chunks = (n_cam + kDenseCamTile - 1) / kDenseCamTile;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > 65535) chunks = 65535;
    if (chunks < 1) chunks = 1;
    int64_t per = (n_cam + chunks - 1) / chunks;
    per = (per + kDenseCamTile - 1) / kDenseCamTile * kDenseCamTile;
    chunks = (n_cam + per - 1) / per;
    *grid = dim3((unsigned)bx, (unsigned)chunks);
    *cams_per_chunk = per;
}

static int dense_check(const char *who, const void *camblk, int64_t n_cam, const void *pts4, int64_t n_pts) {
    if (n_cam < 0 || n_pts < 0 || (n_cam && !camblk) || (n_pts && !pts4)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: bad arguments", who);
    if (n_pts >= ((int64_t)1 << 32)) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: point indices are 32-bit", who);
    if ((n_cam && !aligned16(camblk)) || (n_pts && !aligned16(pts4))) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: camblk/pts4 must be 16-byte aligned", who);
    return C2B_OK;
}

int c2b_visibility_dense_count(const double *camblk, int64_t n_cam, const double *pts4, int64_t n_pts, double max_dist,
                               uint32_t *tile_counts, uint64_t *cam_total, uint64_t *row_ptr, void *stream) {
    C2B_API_BEGIN
    int rc = dense_check("visibility_dense_count", camblk, n_cam, pts4, n_pts);
    if (rc) return rc;
    if (!row_ptr) return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_dense_count: row_ptr is NULL");
    const int64_t n_tiles = c2b_visibility_dense_tiles(n_pts);
    if (!n_cam || !n_tiles) {
        HIP_TRY(hipMemsetAsync(row_ptr, 0, sizeof(uint64_t) * (size_t)(n_cam + 1), S(stream)));
        return C2B_OK;
    }
    if (!tile_counts || !cam_total) return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_dense_count: NULL scratch");
    dim3 grid;
    int64_t per;
    dense_grid(n_cam, n_tiles, &grid, &per);
    // the count pass writes non-empty (camera, tile) cells only
    HIP_TRY(hipMemsetAsync(tile_counts, 0, sizeof(uint32_t) * (size_t)n_cam * (size_t)n_tiles, S(stream)));
    hipLaunchKernelGGL(k_visibility_dense<false>, grid, dim3(kDenseWPB * 64), 0, S(stream), camblk, n_cam, per,
                       reinterpret_cast<const double4 *>(pts4), n_pts, n_tiles, max_dist, tile_counts,
                       (const uint64_t *)nullptr, (uint32_t *)nullptr, (double2 *)nullptr);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dense_row_scan, dim3((unsigned)n_cam), dim3(256), 0, S(stream), tile_counts, n_tiles, cam_total);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dense_cam_scan, dim3(1), dim3(256), 0, S(stream), (const uint64_t *)cam_total, n_cam, row_ptr);
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_dense_count")
}

int c2b_visibility_dense_fill(const double *camblk, int64_t n_cam, const double *pts4, int64_t n_pts, double max_dist,
                              const uint32_t *tile_offsets, const uint64_t *row_ptr, uint32_t *pt_idx, double *uv,
                              void *stream) {
    C2B_API_BEGIN
    int rc = dense_check("visibility_dense_fill", camblk, n_cam, pts4, n_pts);
    if (rc) return rc;
    const int64_t n_tiles = c2b_visibility_dense_tiles(n_pts);
    if (!n_cam || !n_tiles) return C2B_OK;
    if (!tile_offsets || !row_ptr || !pt_idx || !uv || !aligned16(uv))
        return fail(C2B_ERR_INVALID_ARGUMENT, "visibility_dense_fill: NULL/misaligned buffer");
    dim3 grid;
    int64_t per;
    dense_grid(n_cam, n_tiles, &grid, &per);
    hipLaunchKernelGGL(k_visibility_dense<true>, grid, dim3(kDenseWPB * 64), 0, S(stream), camblk, n_cam, per,
                       reinterpret_cast<const double4 *>(pts4), n_pts, n_tiles, max_dist,
                       const_cast<uint32_t *>(tile_offsets), row_ptr, pt_idx, reinterpret_cast<double2 *>(uv));
    LAUNCH_CHECK();
    return C2B_OK;
    C2B_API_END("visibility_dense_fill")
}

/* ------------------------------- host-side generator pieces -------------------------- */

int c2b_synthetic_grid_sizes(int64_t cpb, int64_t ppb, int64_t blocks, int64_t *n_cam, int64_t *n_pts) {
    C2B_API_BEGIN
    if (cpb < 0 || ppb < 0 || blocks < 0 || !n_cam || !n_pts)
        return fail(C2B_ERR_INVALID_ARGUMENT, "synthetic_grid_sizes: bad arguments");
    c2b_host::grid_sizes(cpb, ppb, blocks, n_cam, n_pts);
    return C2B_OK;
    C2B_API_END("synthetic_grid_sizes")
}

int c2b_synthetic_grid_layout(int64_t cpb, int64_t ppb, int64_t blocks, double block_length, double block_inset,
                              double camera_height, double point_height, double *cam_pos3, double *cam_dir9,
                              double *pts3) {
    C2B_API_BEGIN
    if (cpb < 0 || ppb < 0 || blocks < 0 || !cam_pos3 || !cam_dir9 || !pts3)
        return fail(C2B_ERR_INVALID_ARGUMENT, "synthetic_grid_layout: bad arguments");
    // assert!(block_inset * 2. < block_length, ...), src/synthetic.rs:177
    if (!(block_inset * 2.0 < block_length))
        return fail(C2B_ERR_INVALID_ARGUMENT,
                    "Block inset (%g) must be less than half the block length (%g), to not violate physical constraints.",
                    block_inset, block_length);
    c2b_host::grid_layout(cpb, ppb, blocks, block_length, block_inset, camera_height, point_height, cam_pos3,
                          cam_dir9, pts3);
    return C2B_OK;
    C2B_API_END("synthetic_grid_layout")
}

int c2b_synthetic_line_layout(int64_t n_cam, int64_t n_pts, double length, double point_offset, double camera_height,
                              double point_height, double *cam_pos3, double *cam_dir9, double *pts3) {
    C2B_API_BEGIN
    if (n_cam < 0 || n_pts < 0 || (n_cam && (!cam_pos3 || !cam_dir9)) || (n_pts && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "synthetic_line_layout: bad arguments");
    c2b_host::line_layout(n_cam, n_pts, length, point_offset, camera_height, point_height, cam_pos3, cam_dir9, pts3);
    return C2B_OK;
    C2B_API_END("synthetic_line_layout")
}

struct c2b_pairs {
    c2b_host::Pairs v;
};

int c2b_candidate_pairs(const double *centers3, int64_t n_cam, const double *pts3, int64_t n_pts, double max_dist,
                        int64_t cam_lo, int64_t cam_hi, int occlusion, double block_length, double block_inset,
                        int n_threads, c2b_pairs **out) {
    C2B_API_BEGIN
    if (!out) return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: out is NULL");
    *out = nullptr;
    if (n_cam < 0 || n_pts < 0 || cam_lo < 0 || cam_hi > n_cam || cam_lo > cam_hi || (n_cam && !centers3) ||
        (n_pts && !pts3) || !(max_dist >= 0.0))
        return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: bad arguments");
    if (n_cam >= ((int64_t)1 << 32) || n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: indices are 32-bit");
    if (occlusion && !(block_length > 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: block_length must be > 0");
    c2b_pairs *p = new (std::nothrow) c2b_pairs();
    if (!p) return fail(C2B_ERR_OOM, "candidate_pairs: host allocation failed");
    try {
        c2b_host::candidate_pairs(centers3, pts3, n_pts, max_dist, cam_lo, cam_hi, occlusion != 0, block_length,
                                  block_inset, n_threads, &p->v);
    } catch (const std::bad_alloc &) {
        delete p;
        return fail(C2B_ERR_OOM, "candidate_pairs: out of host memory");
    } catch (const std::exception &e) {
        delete p;
        return fail(C2B_ERR_INVALID_ARGUMENT, "candidate_pairs: %s", e.what());
    }
    *out = p;
    return C2B_OK;
    C2B_API_END("candidate_pairs")
}

int64_t c2b_pairs_count(const c2b_pairs *p) { return p ? (int64_t)p->v.cam.size() : 0; }
const uint32_t *c2b_pairs_cam_idx(const c2b_pairs *p) { return p ? p->v.cam.data() : nullptr; }
const uint32_t *c2b_pairs_pt_idx(const c2b_pairs *p) { return p ? p->v.pt.data() : nullptr; }
void c2b_pairs_free(c2b_pairs *p) { delete p; }

/* ---- mesh generator, host side ---- */
struct c2b_obj {
    std::vector<c2b_host::ObjModel> models;
};

int c2b_obj_load(const char *path, c2b_obj **out) {
    C2B_API_BEGIN
    if (!path || !out) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_load: bad arguments");
    *out = nullptr;
    c2b_obj *o = new (std::nothrow) c2b_obj();
    if (!o) return fail(C2B_ERR_OOM, "obj_load: host allocation failed");
    std::string err;
    bool ok = false;
    try {
        ok = c2b_host::load_obj(path, o->models, &err);
    } catch (const std::bad_alloc &) {
        err = "out of host memory";
    }
    if (!ok) { delete o; return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str()); }
    *out = o;
    return C2B_OK;
    C2B_API_END("obj_load")
}

int64_t c2b_obj_model_count(const c2b_obj *o) { return o ? (int64_t)o->models.size() : 0; }

const char *c2b_obj_model_name(const c2b_obj *o, int64_t m) {
    return (o && m >= 0 && m < (int64_t)o->models.size()) ? o->models[(size_t)m].name.c_str() : nullptr;
}

int c2b_obj_model_sizes(const c2b_obj *o, int64_t m, int64_t *n_positions, int64_t *n_indices, int *is_lines) {
    C2B_API_BEGIN
    if (!o || m < 0 || m >= (int64_t)o->models.size()) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_model_sizes: bad model index");
    const c2b_host::ObjModel &mod = o->models[(size_t)m];
    if (n_positions) *n_positions = (int64_t)(mod.positions.size() / 3);
    if (n_indices) *n_indices = (int64_t)mod.indices.size();
    if (is_lines) *is_lines = mod.lines ? 1 : 0;
    return C2B_OK;
    C2B_API_END("obj_model_sizes")
}

int c2b_obj_model_copy(const c2b_obj *o, int64_t m, float *positions3, uint32_t *indices) {
    C2B_API_BEGIN
    if (!o || m < 0 || m >= (int64_t)o->models.size()) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_model_copy: bad model index");
    const c2b_host::ObjModel &mod = o->models[(size_t)m];
    if (positions3) std::copy(mod.positions.begin(), mod.positions.end(), positions3);
    if (indices) std::copy(mod.indices.begin(), mod.indices.end(), indices);
    return C2B_OK;
    C2B_API_END("obj_model_copy")
}

int c2b_obj_move_to_origin(c2b_obj *o, int64_t skip_model) {
    C2B_API_BEGIN
    if (!o) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_move_to_origin: obj is NULL");
    c2b_host::move_to_origin(o->models, skip_model);
    return C2B_OK;
    C2B_API_END("obj_move_to_origin")
}

int c2b_obj_triangles(const c2b_obj *o, int64_t skip_model, float *tri9, int64_t *n_tri) {
    C2B_API_BEGIN
    if (!o || !n_tri) return fail(C2B_ERR_INVALID_ARGUMENT, "obj_triangles: bad arguments");
    std::vector<c2b_host::ObjModel> use;
    for (int64_t m = 0; m < (int64_t)o->models.size(); ++m)
        if (m != skip_model) use.push_back(o->models[(size_t)m]);
    std::vector<float> t;
    c2b_host::triangles_of(use, t);
    *n_tri = (int64_t)(t.size() / 9);
    if (tri9) std::copy(t.begin(), t.end(), tri9);
    return C2B_OK;
    C2B_API_END("obj_triangles")
}

void c2b_obj_free(c2b_obj *o) { delete o; }

int c2b_generate_cameras_path(const c2b_obj *o, int64_t path_model, int64_t num_cameras, double step_size, uint64_t seed,
                              double *cam_pos3, double *cam_dir9) {
    C2B_API_BEGIN
    if (!o || path_model < 0 || path_model >= (int64_t)o->models.size() || num_cameras < 0 || (num_cameras && (!cam_pos3 || !cam_dir9)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "generate_cameras_path: bad arguments");
    const c2b_host::ObjModel &path = o->models[(size_t)path_model];
    if (!path.lines) return fail(C2B_ERR_INVALID_ARGUMENT, "generate_cameras_path: model '%s' is not a polyline", path.name.c_str());
    c2b_host::CameraSamples cs;
    std::string err;
    double total = 0;
    const bool ok = step_size <= 0.0 ? c2b_host::cameras_path(path, num_cameras, seed, cs, &err)
                                     : c2b_host::cameras_path_step(path, num_cameras, step_size, cs, &err, &total);
    if (!ok) return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    std::copy(cs.pos.begin(), cs.pos.end(), cam_pos3);
    std::copy(cs.dir.begin(), cs.dir.end(), cam_dir9);
    return C2B_OK;
    C2B_API_END("generate_cameras_path")
}

int c2b_generate_cameras_poisson(const float *tri9, int64_t n_tri, int64_t num_points, double height, double ground,
                                 uint64_t seed, int64_t capacity, double *cam_pos3, double *cam_dir9, int64_t *n_out) {
    C2B_API_BEGIN
    if (!tri9 || n_tri <= 0 || num_points < 0 || capacity < 0 || !n_out || (capacity && (!cam_pos3 || !cam_dir9)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "generate_cameras_poisson: bad arguments");
    const std::vector<float> tri(tri9, tri9 + 9 * n_tri);
    c2b_host::CameraSamples cs;
    c2b_host::cameras_poisson(tri, num_points, height, ground, seed, cs);
    const int64_t n = std::min<int64_t>((int64_t)cs.size(), capacity);
    if (n) {
        std::copy(cs.pos.begin(), cs.pos.begin() + 3 * n, cam_pos3);
        std::copy(cs.dir.begin(), cs.dir.begin() + 9 * n, cam_dir9);
    }
    *n_out = (int64_t)cs.size();
    return C2B_OK;
    C2B_API_END("generate_cameras_poisson")
}

int c2b_modify_intrinsics(double *cams15, int64_t n_cam, const double start[3], const double end[3], uint64_t seed) {
    C2B_API_BEGIN
    if (n_cam < 0 || (n_cam && !cams15) || !start || !end) return fail(C2B_ERR_INVALID_ARGUMENT, "modify_intrinsics: bad arguments");
    c2b_host::modify_intrinsics(cams15, n_cam, start, end, seed);
    return C2B_OK;
    C2B_API_END("modify_intrinsics")
}

int c2b_generate_world_points(const float *tri9, int64_t n_tri, const double *centers3, int64_t n_cam, int64_t num_points,
                              double max_dist, uint64_t seed, double *pts3, int64_t *n_out) {
    C2B_API_BEGIN
    if (!tri9 || n_tri < 0 || n_cam < 0 || (n_cam && !centers3) || num_points < 0 || !n_out || (num_points && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "generate_world_points: bad arguments");
    const std::vector<float> tri(tri9, tri9 + 9 * n_tri);
    std::vector<double> pts;
    std::string err;
    if (!c2b_host::world_points_uniform(tri, centers3, n_cam, num_points, max_dist, seed, pts, &err))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    std::copy(pts.begin(), pts.end(), pts3);
    *n_out = (int64_t)(pts.size() / 3);
    return C2B_OK;
    C2B_API_END("generate_world_points")
}

static int cull_host(int mode, int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3, uint64_t *row_ptr,
                     uint64_t *pt_idx, double *uv, int faithful) {
    C2B_API_BEGIN
    if (!n_cam || !n_pts || !row_ptr || *n_cam < 0 || *n_pts < 0 || cam_stride < 0 || (*n_cam && cam_stride && !cams) ||
        (*n_pts && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "cull: bad arguments");
    const int64_t n_obs = (int64_t)row_ptr[*n_cam];
    if (n_obs && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "cull: NULL observations");
    for (int64_t o = 0; o < n_obs; ++o)
        if (pt_idx[o] >= (uint64_t)*n_pts) return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "cull: point index out of range");
    try {
        c2b_host::Graph g;
        g.n_cam = *n_cam; g.n_pts = *n_pts; g.stride = cam_stride;
        g.cams.assign(cams, cams + (size_t)*n_cam * cam_stride);
        g.pts.assign(pts3, pts3 + (size_t)*n_pts * 3);
        g.row_ptr.assign(row_ptr, row_ptr + *n_cam + 1);
        g.pt_idx.assign(pt_idx, pt_idx + n_obs);
        g.uv.assign(uv, uv + 2 * n_obs);
        const c2b_host::Graph c = c2b_host::cull(g, faithful != 0, mode);
        std::copy(c.cams.begin(), c.cams.end(), cams);
        std::copy(c.pts.begin(), c.pts.end(), pts3);
        std::copy(c.row_ptr.begin(), c.row_ptr.end(), row_ptr);
        std::copy(c.pt_idx.begin(), c.pt_idx.end(), pt_idx);
        std::copy(c.uv.begin(), c.uv.end(), uv);
        *n_cam = c.n_cam;
        *n_pts = c.n_pts;
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "cull: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("cull_host")
}

int c2b_cull(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3, uint64_t *row_ptr,
             uint64_t *pt_idx, double *uv, int faithful) {
    C2B_API_BEGIN
    return cull_host(0, n_cam, cams, cam_stride, n_pts, pts3, row_ptr, pt_idx, uv, faithful);
    C2B_API_END("cull")
}
int c2b_largest_connected_component(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3,
                                    uint64_t *row_ptr, uint64_t *pt_idx, double *uv, int faithful) {
    C2B_API_BEGIN
    return cull_host(1, n_cam, cams, cam_stride, n_pts, pts3, row_ptr, pt_idx, uv, faithful);
    C2B_API_END("largest_connected_component")
}
int c2b_remove_singletons(int64_t *n_cam, double *cams, int cam_stride, int64_t *n_pts, double *pts3, uint64_t *row_ptr,
                          uint64_t *pt_idx, double *uv) {
    C2B_API_BEGIN
    return cull_host(2, n_cam, cams, cam_stride, n_pts, pts3, row_ptr, pt_idx, uv, 1);
    C2B_API_END("remove_singletons")
}

/* ---- index-corruption noise, host side ---- */
static int check_csr(const char *who, int64_t n_cam, const uint64_t *row_ptr) {
    if (n_cam < 0 || !row_ptr) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: bad arguments", who);
    if (row_ptr[0] != 0) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: row_ptr[0] != 0", who);
    for (int64_t c = 0; c < n_cam; ++c)
        if (row_ptr[c + 1] < row_ptr[c]) return fail(C2B_ERR_INVALID_ARGUMENT, "%s: row_ptr not monotone at camera %lld", who, (long long)c);
    return C2B_OK;
}

int c2b_add_incorrect_correspondences(int64_t n_cam, const uint64_t *row_ptr, uint64_t *pt_idx, const double *uv,
                                      double mismatch_chance, uint64_t seed) {
    C2B_API_BEGIN
    int rc = check_csr("add_incorrect_correspondences", n_cam, row_ptr);
    if (rc) return rc;
    if (row_ptr[n_cam] && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "add_incorrect_correspondences: NULL observations");
    std::string err;
    try {
        if (!c2b_host::add_incorrect_correspondences(n_cam, row_ptr, pt_idx, uv, mismatch_chance, seed, &err))
            return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "add_incorrect_correspondences: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("add_incorrect_correspondences")
}

int c2b_drop_features(int64_t n_cam, uint64_t *row_ptr, uint64_t *pt_idx, double *uv, double keep_fraction, uint64_t seed) {
    C2B_API_BEGIN
    int rc = check_csr("drop_features", n_cam, row_ptr);
    if (rc) return rc;
    if (row_ptr[n_cam] && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "drop_features: NULL observations");
    if (keep_fraction != keep_fraction) return fail(C2B_ERR_INVALID_ARGUMENT, "drop_features: keep_fraction is NaN");
    try {
        c2b_host::drop_features(n_cam, row_ptr, pt_idx, uv, keep_fraction, seed);
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "drop_features: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("drop_features")
}

int c2b_split_landmarks(int64_t *n_pts, double *pts3, int64_t pts_capacity, int64_t n_obs, uint64_t *pt_idx,
                        double split_fraction, uint64_t seed) {
    C2B_API_BEGIN
    if (!n_pts || *n_pts < 0 || n_obs < 0 || (*n_pts && !pts3) || (n_obs && !pt_idx) || split_fraction != split_fraction)
        return fail(C2B_ERR_INVALID_ARGUMENT, "split_landmarks: bad arguments");
    const uint64_t n = std::min<uint64_t>(c2b_host::fraction_of(split_fraction, (uint64_t)*n_pts), (uint64_t)*n_pts);
    if (pts_capacity < *n_pts + (int64_t)n)
        return fail(C2B_ERR_INVALID_ARGUMENT, "split_landmarks: pts3 holds %lld rows, %lld needed", (long long)pts_capacity,
                    (long long)(*n_pts + (int64_t)n));
    for (int64_t o = 0; o < n_obs; ++o)
        if (pt_idx[o] >= (uint64_t)*n_pts) return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "split_landmarks: point index out of range");
    try {
        *n_pts = c2b_host::split_landmarks(*n_pts, pts3, n_obs, pt_idx, split_fraction, seed);
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "split_landmarks: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("split_landmarks")
}

int c2b_join_landmarks(int64_t n_pts, const double *pts3, int64_t n_obs, uint64_t *pt_idx, double join_fraction, uint64_t seed) {
    C2B_API_BEGIN
    if (n_pts < 0 || n_obs < 0 || (n_pts && !pts3) || (n_obs && !pt_idx) || join_fraction != join_fraction)
        return fail(C2B_ERR_INVALID_ARGUMENT, "join_landmarks: bad arguments");
    for (int64_t o = 0; o < n_obs; ++o)
        if (pt_idx[o] >= (uint64_t)n_pts) return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "join_landmarks: point index out of range");
    std::string err;
    try {
        if (!c2b_host::join_landmarks(n_pts, pts3, n_obs, pt_idx, join_fraction, seed, &err))
            return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    } catch (const std::bad_alloc &) {
        return fail(C2B_ERR_OOM, "join_landmarks: out of host memory");
    }
    return C2B_OK;
    C2B_API_END("join_landmarks")
}

struct c2b_balfile {
    c2b_host::Graph g;
};

// format: 0 text (from_file_text), 1 binary (from_file_binary), -1 by extension (from_file)
static int bal_format(const char *path, int format, bool *binary) {
    if (format == 0 || format == 1) { *binary = format == 1; return C2B_OK; }
    const std::string ext = c2b_host::extension(path);
    if (ext.empty()) return fail(C2B_ERR_INVALID_ARGUMENT, "file does not have an extension");
    if (ext != "bal" && ext != "bbal") return fail(C2B_ERR_INVALID_ARGUMENT, "unknown file extension %s", ext.c_str());
    *binary = ext == "bbal";
    return C2B_OK;
}

int c2b_bal_read_as(const char *path, int format, c2b_balfile **out) {
    C2B_API_BEGIN
    if (!path || !out) return fail(C2B_ERR_INVALID_ARGUMENT, "bal_read: bad arguments");
    *out = nullptr;
    bool binary = false;
    int rc = bal_format(path, format, &binary);
    if (rc) return rc;
    const std::string ext = binary ? "bbal" : "bal";
    c2b_balfile *f = new (std::nothrow) c2b_balfile();
    if (!f) return fail(C2B_ERR_OOM, "bal_read: host allocation failed");
    std::string err;
    bool ok = false;
    try {
        ok = ext == "bal" ? c2b_host::read_text(path, f->g, &err) : c2b_host::read_binary(path, f->g, &err);
    } catch (const std::bad_alloc &) {
        err = "out of host memory";
    }
    if (!ok) {
        delete f;
        const bool range = err.find("assertion failed") != std::string::npos || err.find("out of range") != std::string::npos;
        return fail(range ? C2B_ERR_INDEX_OUT_OF_RANGE : C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    }
    *out = f;
    return C2B_OK;
    C2B_API_END("bal_read_as")
}

int c2b_bal_read(const char *path, c2b_balfile **out) { return c2b_bal_read_as(path, -1, out); }

int c2b_bal_sizes(const c2b_balfile *f, int64_t *n_cam, int64_t *n_pts, int64_t *n_obs) {
    C2B_API_BEGIN
    if (!f) return fail(C2B_ERR_INVALID_ARGUMENT, "bal_sizes: file is NULL");
    if (n_cam) *n_cam = f->g.n_cam;
    if (n_pts) *n_pts = f->g.n_pts;
    if (n_obs) *n_obs = f->g.n_obs();
    return C2B_OK;
    C2B_API_END("bal_sizes")
}

int c2b_bal_copy(const c2b_balfile *f, double *bal9, double *pts3, uint64_t *row_ptr, uint64_t *pt_idx, double *uv) {
    C2B_API_BEGIN
    if (!f) return fail(C2B_ERR_INVALID_ARGUMENT, "bal_copy: file is NULL");
    if (bal9) std::copy(f->g.cams.begin(), f->g.cams.end(), bal9);
    if (pts3) std::copy(f->g.pts.begin(), f->g.pts.end(), pts3);
    if (row_ptr) std::copy(f->g.row_ptr.begin(), f->g.row_ptr.end(), row_ptr);
    if (pt_idx) std::copy(f->g.pt_idx.begin(), f->g.pt_idx.end(), pt_idx);
    if (uv) std::copy(f->g.uv.begin(), f->g.uv.end(), uv);
    return C2B_OK;
    C2B_API_END("bal_copy")
}

void c2b_bal_close(c2b_balfile *f) { delete f; }

int c2b_bal_write_as(const char *path, int format, int64_t n_cam, const double *bal9, int64_t n_pts, const double *pts3,
                     const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    C2B_API_BEGIN
    if (!path || n_cam < 0 || n_pts < 0 || !row_ptr || (n_cam && !bal9) || (n_pts && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "bal_write: bad arguments");
    bool binary = false;
    int rc = bal_format(path, format, &binary);
    if (rc) return rc;
    const std::string ext = binary ? "bbal" : "bal";
    const int64_t n_obs = (int64_t)row_ptr[n_cam];
    if (n_obs && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "bal_write: NULL observations");
    std::string err;
    bool ok = false;
    try {
        c2b_host::Graph g;
        g.n_cam = n_cam; g.n_pts = n_pts; g.stride = 9;
        g.cams.assign(bal9, bal9 + (size_t)n_cam * 9);
        g.pts.assign(pts3, pts3 + (size_t)n_pts * 3);
        g.row_ptr.assign(row_ptr, row_ptr + n_cam + 1);
        g.pt_idx.assign(pt_idx, pt_idx + n_obs);
        g.uv.assign(uv, uv + 2 * n_obs);
        ok = ext == "bal" ? c2b_host::write_text(path, g, &err) : c2b_host::write_binary(path, g, &err);
    } catch (const std::bad_alloc &) {
        err = "out of host memory";
    }
    if (!ok) return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    return C2B_OK;
    C2B_API_END("bal_write_as")
}

int c2b_bal_write(const char *path, int64_t n_cam, const double *bal9, int64_t n_pts, const double *pts3,
                  const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    C2B_API_BEGIN
    return c2b_bal_write_as(path, -1, n_cam, bal9, n_pts, pts3, row_ptr, pt_idx, uv);
    C2B_API_END("bal_write")
}

int c2b_ply_write(const char *path, int64_t n_cam, const double *centers3, int64_t n_pts, const double *pts3,
                  const uint64_t *row_ptr, const uint64_t *pt_idx) {
    C2B_API_BEGIN
    if (!path || n_cam < 0 || n_pts < 0 || (n_cam && (!centers3 || !row_ptr)) || (n_pts && !pts3))
        return fail(C2B_ERR_INVALID_ARGUMENT, "ply_write: bad arguments");
    if (n_cam && row_ptr[n_cam] && !pt_idx) return fail(C2B_ERR_INVALID_ARGUMENT, "ply_write: pt_idx is NULL");
    std::string err;
    if (!c2b_host::write_ply(path, n_cam, centers3, n_pts, pts3, row_ptr, pt_idx, &err))
        return fail(C2B_ERR_INVALID_ARGUMENT, "%s", err.c_str());
    return C2B_OK;
    C2B_API_END("ply_write")
}

/* ------------------------------- level 1 --------------------------------------------- */

struct c2b_problem {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n_cam = 0, n_pts = 0, n_obs = 0;
    double *cam15 = nullptr, *bal9 = nullptr, *camblk = nullptr, *pts4 = nullptr, *uv = nullptr;
    uint32_t *cam_idx = nullptr, *pt_idx = nullptr;
    void *ws = nullptr;
    double *stats = nullptr, *scalar = nullptr;
    // this problem as ONE SHARD of a larger one (c2b_problem_set_shard): its cameras are [shard_cam_base, + n_cam) of
    // shard_n_cam_global (< 0: not a shard), its first observation is observation shard_obs_base of the whole list
    int64_t shard_cam_base = 0, shard_n_cam_global = -1, shard_obs_base = 0;
    bool bal_valid = false;     // bal9 still describes the cameras (no mutation since upload_bal)
    bool blk_valid = false;     // camblk matches cam15 (and bal_valid mode)
    // the row structure of the observation list for the *_rows launchers, rebuilt on demand after the list changed
    uint64_t *rows_ptr = nullptr;
    void *rows_tiles = nullptr;
    bool rows_valid = false;
    uint32_t *dense_pt = nullptr;   // survivors of the last dense visibility sweep
    double *dense_uv = nullptr;
    uint64_t *dense_row = nullptr;  // its CSR row pointer [n_cam + 1], kept for the occlusion filter
    int64_t dense_n = 0;
    // residual + Jacobian to host buffers: a ring of chunk-sized device buffers, a copy stream, per-slot events
    static constexpr int kJacSlots = 3;
    static constexpr int64_t kJacChunk = 256 * 1024;       // observations per chunk (53 MB of results)
    double *jac_ring = nullptr;                            // kJacSlots x kJacChunk x 26 doubles
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_done[kJacSlots] = {nullptr, nullptr, nullptr}, ev_free[kJacSlots] = {nullptr, nullptr, nullptr};
};

static void free_dense(c2b_problem *p) {
    if (p->dense_pt) (void)hipFree(p->dense_pt);
    if (p->dense_uv) (void)hipFree(p->dense_uv);
    if (p->dense_row) (void)hipFree(p->dense_row);
    p->dense_pt = nullptr; p->dense_uv = nullptr; p->dense_row = nullptr; p->dense_n = 0;
}

// the observation list changed (upload, cull, adopted visibility): its row structure is rebuilt by the next user
static void drop_rows(c2b_problem *p) {
    if (p->rows_ptr) (void)hipFree(p->rows_ptr);
    if (p->rows_tiles) (void)hipFree(p->rows_tiles);
    p->rows_ptr = nullptr; p->rows_tiles = nullptr; p->rows_valid = false;
}

static void free_buffers(c2b_problem *p) {
    void *ptrs[] = {p->cam15, p->bal9, p->camblk, p->pts4, p->uv, p->cam_idx, p->pt_idx, p->ws, p->stats, p->scalar, p->jac_ring};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    p->jac_ring = nullptr;
    free_dense(p);
    drop_rows(p);
    p->cam15 = p->bal9 = p->camblk = p->pts4 = p->uv = nullptr;
    p->cam_idx = p->pt_idx = nullptr;
    p->ws = nullptr; p->stats = p->scalar = nullptr;
    p->n_cam = p->n_pts = p->n_obs = 0;
    p->bal_valid = p->blk_valid = false;
}

int c2b_problem_create(int device, c2b_problem **out) {
    C2B_API_BEGIN
    if (!out) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(C2B_ERR_NO_DEVICE, "problem_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_create: device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    c2b_problem *p = new (std::nothrow) c2b_problem();
    if (!p) return fail(C2B_ERR_OOM, "problem_create: host allocation failed");
    p->device = device;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete p; return fail(C2B_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    *out = p;
    return C2B_OK;
    C2B_API_END("problem_create")
}

void c2b_problem_destroy(c2b_problem *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    free_buffers(p);
    for (int k = 0; k < c2b_problem::kJacSlots; ++k) {
        if (p->ev_done[k]) (void)hipEventDestroy(p->ev_done[k]);
        if (p->ev_free[k]) (void)hipEventDestroy(p->ev_free[k]);
    }
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

static int ensure_camblk(c2b_problem *p) {
    if (p->blk_valid) return C2B_OK;
    int rc = p->bal_valid ? c2b_cameras_prepare_bal(p->bal9, p->n_cam, p->camblk, p->stream)
                          : c2b_cameras_prepare_state(p->cam15, p->n_cam, p->camblk, p->stream);
    if (rc) return rc;
    p->blk_valid = true;
    return C2B_OK;
}

// the resident arrays of a problem with these sizes (whatever it held before is freed); contents undefined
static int alloc_problem(c2b_problem *p, int64_t n_cam, int64_t n_pts, int64_t n_obs) {
    HIP_TRY(hipSetDevice(p->device));
    free_buffers(p);
    auto dalloc = [&](void **q, size_t bytes) -> hipError_t { return hipMalloc(q, bytes ? bytes : 16); };
    HIP_TRY(dalloc((void **)&p->cam15, sizeof(double) * 15 * n_cam));
    HIP_TRY(dalloc((void **)&p->bal9, sizeof(double) * 9 * n_cam));
    HIP_TRY(dalloc((void **)&p->camblk, sizeof(double) * kCamBlk * n_cam));
    HIP_TRY(dalloc((void **)&p->pts4, sizeof(double) * 4 * n_pts));
    HIP_TRY(dalloc((void **)&p->uv, sizeof(double) * 2 * n_obs));
    HIP_TRY(dalloc((void **)&p->cam_idx, sizeof(uint32_t) * n_obs));
    HIP_TRY(dalloc((void **)&p->pt_idx, sizeof(uint32_t) * n_obs));
    HIP_TRY(dalloc(&p->ws, (size_t)c2b_workspace_bytes(n_obs)));
    if (int rc = c2b_workspace_init(p->ws, p->stream)) return rc;
    HIP_TRY(dalloc((void **)&p->stats, sizeof(double) * C2B_STATS_DOUBLES));
    HIP_TRY(dalloc((void **)&p->scalar, sizeof(double) * 2));
    p->n_cam = n_cam; p->n_pts = n_pts; p->n_obs = n_obs;
    return C2B_OK;
}

static int upload_common(c2b_problem *p, int64_t n_cam, const double *cams, bool is_bal, int64_t n_pts,
                         const double *pts3, const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: problem is NULL");
    if (n_cam < 0 || n_pts < 0 || (n_cam && !cams) || (n_pts && !pts3) || !row_ptr)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: bad arguments");
    if (n_cam >= ((int64_t)1 << 32) || n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: device indices are 32-bit");
    // assert!(cams.len() == obs.len()) is structural here; row_ptr must be a monotone prefix
    if (row_ptr[0] != 0) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: row_ptr[0] != 0");
    for (int64_t c = 0; c < n_cam; ++c)
        if (row_ptr[c + 1] < row_ptr[c]) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: row_ptr not monotone at camera %lld", (long long)c);
    const int64_t n_obs = (int64_t)row_ptr[n_cam];
    if (n_obs && (!pt_idx || !uv)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_upload: NULL observations");
    std::vector<uint32_t> pi32((size_t)n_obs);
    for (int64_t o = 0; o < n_obs; ++o) {
        // assert!(ci < &points.len()), src/baproblem.rs:368
        if (pt_idx[o] >= (uint64_t)n_pts)
            return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "problem_upload: observation %lld refers to point %llu >= %lld",
                        (long long)o, (unsigned long long)pt_idx[o], (long long)n_pts);
        pi32[(size_t)o] = (uint32_t)pt_idx[o];
    }
    if (int rc = alloc_problem(p, n_cam, n_pts, n_obs)) return rc;
    auto dalloc = [&](void **q, size_t bytes) -> hipError_t { return hipMalloc(q, bytes ? bytes : 16); };

    // staging through temporary device buffers (row_ptr, packed points)
    uint64_t *d_row = nullptr;
    double *d_p3 = nullptr;
    HIP_TRY(dalloc((void **)&d_row, sizeof(uint64_t) * (n_cam + 1)));
    hipError_t e = dalloc((void **)&d_p3, sizeof(double) * 3 * n_pts);
    if (e != hipSuccess) { (void)hipFree(d_row); return fail(C2B_ERR_OOM, "problem_upload: %s", hipGetErrorString(e)); }
    int rc = C2B_OK;
    do {
#define UP_TRY(expr) { hipError_t e2 = (expr); if (e2 != hipSuccess) { rc = fail(C2B_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e2)); break; } }
        UP_TRY(hipMemcpyAsync(d_row, row_ptr, sizeof(uint64_t) * (n_cam + 1), hipMemcpyHostToDevice, p->stream));
        if (n_pts) UP_TRY(hipMemcpyAsync(d_p3, pts3, sizeof(double) * 3 * n_pts, hipMemcpyHostToDevice, p->stream));
        if (n_obs) {
            UP_TRY(hipMemcpyAsync(p->pt_idx, pi32.data(), sizeof(uint32_t) * n_obs, hipMemcpyHostToDevice, p->stream));
            UP_TRY(hipMemcpyAsync(p->uv, uv, sizeof(double) * 2 * n_obs, hipMemcpyHostToDevice, p->stream));
        }
        if (is_bal) {
            if (n_cam) UP_TRY(hipMemcpyAsync(p->bal9, cams, sizeof(double) * 9 * n_cam, hipMemcpyHostToDevice, p->stream));
            if ((rc = c2b_cameras_from_bal(p->bal9, n_cam, p->cam15, p->stream))) break;
        } else {
            if (n_cam) UP_TRY(hipMemcpyAsync(p->cam15, cams, sizeof(double) * 15 * n_cam, hipMemcpyHostToDevice, p->stream));
        }
        if ((rc = c2b_points_pad(d_p3, n_pts, p->pts4, p->stream))) break;
        if ((rc = c2b_expand_rows(d_row, n_cam, 0, n_obs, p->cam_idx, p->stream))) break;
        UP_TRY(hipStreamSynchronize(p->stream));
#undef UP_TRY
    } while (0);
    (void)hipFree(d_row);
    (void)hipFree(d_p3);
    if (rc) return rc;
    p->bal_valid = is_bal;
    p->blk_valid = false;
    return C2B_OK;
}

int c2b_problem_upload(c2b_problem *p, int64_t n_cam, const double *cams15, int64_t n_pts, const double *pts3,
                       const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    C2B_API_BEGIN
    return upload_common(p, n_cam, cams15, false, n_pts, pts3, row_ptr, pt_idx, uv);
    C2B_API_END("problem_upload")
}

int c2b_problem_upload_bal(c2b_problem *p, int64_t n_cam, const double *bal9, int64_t n_pts, const double *pts3,
                           const uint64_t *row_ptr, const uint64_t *pt_idx, const double *uv) {
    C2B_API_BEGIN
    return upload_common(p, n_cam, bal9, true, n_pts, pts3, row_ptr, pt_idx, uv);
    C2B_API_END("problem_upload_bal")
}

// synthetic_grid's / synthetic_line's layout loops (src/synthetic.rs:178-258, :323-344) straight into the resident problem:
// cameras by Camera::from_position_direction, points, no observations yet (the visibility loop adds them).  Entity for
// entity and bit for bit what c2b_synthetic_grid_layout + c2b_problem_from_position_direction + c2b_problem_upload give,
// without 2 x 112 MB crossing PCIe.  The orientations' sines and cosines come from the host's libm like the host
// layout's (Basis3::from_angle_y(Deg(..)), :191-205).
static GridDirs layout_dirs() {
    GridDirs d;
    c2b_host::basis_from_angle_y_deg(-90.0, d.m[0]);
    c2b_host::basis_from_angle_y_deg(90.0, d.m[1]);
    c2b_host::basis_from_angle_y_deg(180.0, d.m[2]);
    const double one[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::copy(one, one + 9, d.m[3]);
    return d;
}

int c2b_problem_synthetic_grid_layout(c2b_problem *p, int64_t cpb, int64_t ppb, int64_t blocks, double block_length,
                                      double block_inset, double camera_height, double point_height) {
    C2B_API_BEGIN
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_grid_layout: problem is NULL");
    if (cpb < 0 || ppb < 0 || blocks < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_grid_layout: bad arguments");
    // assert!(block_inset * 2. < block_length, ...), src/synthetic.rs:177
    if (!(block_inset * 2.0 < block_length))
        return fail(C2B_ERR_INVALID_ARGUMENT,
                    "Block inset (%g) must be less than half the block length (%g), to not violate physical constraints.",
                    block_inset, block_length);
    int64_t n_cam = 0, n_pts = 0;
    c2b_host::grid_sizes(cpb, ppb, blocks, &n_cam, &n_pts);
    if (n_cam >= ((int64_t)1 << 32) || n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_grid_layout: device indices are 32-bit");
    int rc = alloc_problem(p, n_cam, n_pts, 0);
    if (rc) return rc;
    if (n_cam) hipLaunchKernelGGL(k_grid_cameras, dim3(blocks_for(n_cam, 256)), dim3(256), 0, p->stream, n_cam, cpb, blocks, block_length,
                                  camera_height, layout_dirs(), p->cam15);
    if (n_pts) hipLaunchKernelGGL(k_grid_points, dim3(blocks_for(n_pts, 256)), dim3(256), 0, p->stream, n_pts, ppb, blocks, block_length,
                                  block_inset, point_height, reinterpret_cast<double4 *>(p->pts4));
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->bal_valid = false; p->blk_valid = false;
    return C2B_OK;
    C2B_API_END("problem_synthetic_grid_layout")
}

int c2b_problem_synthetic_line_layout(c2b_problem *p, int64_t n_cam, int64_t n_pts, double length, double point_offset,
                                      double camera_height, double point_height) {
    C2B_API_BEGIN
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_line_layout: problem is NULL");
    if (n_cam < 0 || n_pts < 0 || n_cam >= ((int64_t)1 << 32) || n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_synthetic_line_layout: bad arguments");
    int rc = alloc_problem(p, n_cam, n_pts, 0);
    if (rc) return rc;
    const int64_t n = std::max(n_cam, n_pts);
    if (n) hipLaunchKernelGGL(k_line_layout, dim3(blocks_for(n, 256)), dim3(256), 0, p->stream, n_cam, n_pts, length, point_offset,
                              camera_height, point_height, layout_dirs(), p->cam15, reinterpret_cast<double4 *>(p->pts4));
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->bal_valid = false; p->blk_valid = false;
    return C2B_OK;
    C2B_API_END("problem_synthetic_line_layout")
}

int c2b_problem_sizes(const c2b_problem *p, int64_t *n_cam, int64_t *n_pts, int64_t *n_obs) {
    C2B_API_BEGIN
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_sizes: problem is NULL");
    if (n_cam) *n_cam = p->n_cam;
    if (n_pts) *n_pts = p->n_pts;
    if (n_obs) *n_obs = p->n_obs;
    return C2B_OK;
    C2B_API_END("problem_sizes")
}

#define NEED_UPLOADED(p, who)                                                              \
    if (!(p)) return fail(C2B_ERR_INVALID_ARGUMENT, who ": problem is NULL");              \
    if (!(p)->ws) return fail(C2B_ERR_INVALID_ARGUMENT, who ": nothing uploaded");         \
    HIP_TRY(hipSetDevice((p)->device));

int c2b_problem_download(c2b_problem *p, double *cams15, double *pts3, double *uv) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_download");
    if (cams15 && p->n_cam)
        HIP_TRY(hipMemcpyAsync(cams15, p->cam15, sizeof(double) * 15 * p->n_cam, hipMemcpyDeviceToHost, p->stream));
    if (uv && p->n_obs)
        HIP_TRY(hipMemcpyAsync(uv, p->uv, sizeof(double) * 2 * p->n_obs, hipMemcpyDeviceToHost, p->stream));
    double *d_p3 = nullptr;
    if (pts3 && p->n_pts) {
        HIP_TRY(hipMalloc((void **)&d_p3, sizeof(double) * 3 * p->n_pts));
        int rc = c2b_points_unpad(p->pts4, p->n_pts, d_p3, p->stream);
        if (rc) { (void)hipFree(d_p3); return rc; }
        hipError_t e = hipMemcpyAsync(pts3, d_p3, sizeof(double) * 3 * p->n_pts, hipMemcpyDeviceToHost, p->stream);
        if (e != hipSuccess) { (void)hipFree(d_p3); return fail(C2B_ERR_HIP, "download points: %s", hipGetErrorString(e)); }
    }
    hipError_t e = hipStreamSynchronize(p->stream);
    if (d_p3) (void)hipFree(d_p3);
    if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_download: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_download")
}

int c2b_problem_download_bal(c2b_problem *p, double *bal9) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_download_bal");
    if (!bal9) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_download_bal: bal9 is NULL");
    if (!p->n_cam) return C2B_OK;
    if (!p->bal_valid) {
        // to_vec (src/baproblem.rs:189-202) of the current state
        int rc = c2b_cameras_to_bal(p->cam15, p->n_cam, p->bal9, p->stream);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(bal9, p->bal9, sizeof(double) * 9 * p->n_cam, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_download_bal")
}

int c2b_problem_from_position_direction(c2b_problem *p, int64_t n_cam, const double *pos3, const double *dir9,
                                        double *cams15) {
    C2B_API_BEGIN
    if (!p) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_from_position_direction: problem is NULL");
    if (n_cam < 0 || (n_cam && (!pos3 || !dir9 || !cams15)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_from_position_direction: bad arguments");
    if (!n_cam) return C2B_OK;
    HIP_TRY(hipSetDevice(p->device));
    double *d_pos = nullptr, *d_dir = nullptr, *d_cam = nullptr;
    int rc = C2B_OK;
    hipError_t e = hipMalloc((void **)&d_pos, sizeof(double) * 3 * n_cam);
    if (e == hipSuccess) e = hipMalloc((void **)&d_dir, sizeof(double) * 9 * n_cam);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cam, sizeof(double) * 15 * n_cam);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pos, pos3, sizeof(double) * 3 * n_cam, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dir, dir9, sizeof(double) * 9 * n_cam, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        rc = c2b_cameras_from_position_direction(d_pos, d_dir, n_cam, d_cam, p->stream);
        if (!rc) e = hipMemcpyAsync(cams15, d_cam, sizeof(double) * 15 * n_cam, hipMemcpyDeviceToHost, p->stream);
        hipError_t e2 = hipStreamSynchronize(p->stream);
        if (e == hipSuccess) e = e2;
    }
    if (d_pos) (void)hipFree(d_pos);
    if (d_dir) (void)hipFree(d_dir);
    if (d_cam) (void)hipFree(d_cam);
    if (rc) return rc;
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_from_position_direction: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_from_position_direction")
}

int c2b_problem_centers(c2b_problem *p, double *centers3) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_centers");
    if (!p->n_cam) return C2B_OK;
    if (!centers3) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_centers: centers3 is NULL");
    int rc = ensure_camblk(p);
    if (rc) return rc;
    // camblk rows are C2B_CAMBLK_DOUBLES doubles; the center sits at [24..26]
    HIP_TRY(hipMemcpy2DAsync(centers3, 3 * sizeof(double), p->camblk + kCenter, kCamBlk * sizeof(double),
                             3 * sizeof(double), (size_t)p->n_cam, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_centers")
}

// row_ptr (from the camera-major cam_idx) and the tile records of the current observation list
static int ensure_rows(c2b_problem *p) {
    if (p->rows_valid || !p->n_obs) return C2B_OK;
    drop_rows(p);
    HIP_TRY(hipMalloc((void **)&p->rows_ptr, sizeof(uint64_t) * (size_t)(p->n_cam + 1)));
    HIP_TRY(hipMalloc(&p->rows_tiles, (size_t)c2b_rows_tiles_bytes(p->n_obs)));
    hipLaunchKernelGGL(k_rows_from_sorted, dim3(blocks_for(p->n_obs + 1)), dim3(kBlock), 0, p->stream, (const uint32_t *)p->cam_idx,
                       p->n_obs, p->n_cam, p->rows_ptr);
    LAUNCH_CHECK();
    const int rc = c2b_rows_pack(p->rows_ptr, p->n_cam, p->n_obs, p->rows_tiles, p->stream);
    if (rc) return rc;
    p->rows_valid = true;
    return C2B_OK;
}

int c2b_problem_project(c2b_problem *p, double *uv_out) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_project");
    if (!p->n_obs) return C2B_OK;
    if (!uv_out) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_project: uv_out is NULL");
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    double *d_uv = nullptr;
    HIP_TRY(hipMalloc((void **)&d_uv, sizeof(double) * 2 * p->n_obs));
    rc = c2b_project_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->n_obs, d_uv, p->stream);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(uv_out, d_uv, sizeof(double) * 2 * p->n_obs, hipMemcpyDeviceToHost, p->stream);
    hipError_t e2 = hipStreamSynchronize(p->stream);
    (void)hipFree(d_uv);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail(C2B_ERR_HIP, "problem_project: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return C2B_OK;
    C2B_API_END("problem_project")
}

int c2b_problem_total_reprojection_error(c2b_problem *p, double norm, double *out) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_total_reprojection_error");
    if (!out) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_error: out is NULL");
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    rc = c2b_reprojection_error_sum_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv, p->n_obs,
                                         norm, p->ws, p->scalar, p->stream);
    if (rc) return rc;
    double sum = 0.0;
    HIP_TRY(hipMemcpyAsync(&sum, p->scalar, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *out = std::pow(sum, 1.0 / norm);          // .powf(1. / norm), src/baproblem.rs:278
    return C2B_OK;
    C2B_API_END("problem_total_reprojection_error")
}

// The same for a problem that is one SHARD (a contiguous camera range) of a larger one: the local sum, one 8-byte
// all-reduce through the communicator on the problem's stream, then .powf(1/norm) -- every rank returns the global
// error (src/baproblem.rs:265-279 over all shards).  Collective: every rank of the communicator must call it.
int c2b_problem_total_reprojection_error_sharded(c2b_problem *p, c2b_comm *comm, double norm, double *out) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_total_reprojection_error_sharded");
    if (!out || !comm) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_error_sharded: NULL argument");
    if (comm->device != p->device) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_error_sharded: communicator and problem live on different devices");
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    if (p->n_obs > 0)
        rc = c2b_reprojection_error_sum_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv, p->n_obs,
                                             norm, p->ws, p->scalar, p->stream);
    else
        HIP_TRY(hipMemsetAsync(p->scalar, 0, sizeof(double), p->stream));          // an empty shard still takes part
    if (!rc) rc = c2b_comm_all_reduce_sum_f64(comm, p->scalar, 1, p->stream);
    if (rc) return rc;
    double sum = 0.0;
    HIP_TRY(hipMemcpyAsync(&sum, p->scalar, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *out = std::pow(sum, 1.0 / norm);
    return C2B_OK;
    C2B_API_END("problem_total_reprojection_error_sharded")
}

// Both norms run_noise prints (src/bin/city2ba.rs:283-287, 350-354) from ONE pass over the observations.
static int errors_l1_l2_impl(c2b_problem *p, c2b_comm *comm, double *l1, double *l2) {
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    if (p->n_obs > 0)
        rc = c2b_reprojection_error_sums2_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv, p->n_obs,
                                               p->ws, p->scalar, p->stream);
    else
        HIP_TRY(hipMemsetAsync(p->scalar, 0, 2 * sizeof(double), p->stream));      // an empty shard still takes part
    if (!rc && comm) rc = c2b_comm_all_reduce_sum_f64(comm, p->scalar, 2, p->stream);   // ONE 2-element all-reduce
    if (rc) return rc;
    double sums[2] = {0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(sums, p->scalar, 2 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *l1 = std::pow(sums[0], 1.0 / 1.0);        // .powf(1. / norm), src/baproblem.rs:278
    *l2 = std::pow(sums[1], 1.0 / 2.0);
    return C2B_OK;
}

int c2b_problem_total_reprojection_errors_l1_l2(c2b_problem *p, double *l1, double *l2) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_total_reprojection_errors_l1_l2");
    if (!l1 || !l2) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_errors_l1_l2: NULL output");
    return errors_l1_l2_impl(p, nullptr, l1, l2);
    C2B_API_END("problem_total_reprojection_errors_l1_l2")
}

int c2b_problem_total_reprojection_errors_l1_l2_sharded(c2b_problem *p, c2b_comm *comm, double *l1, double *l2) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_total_reprojection_errors_l1_l2_sharded");
    if (!l1 || !l2 || !comm) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_errors_l1_l2_sharded: NULL argument");
    if (comm->device != p->device) return fail(C2B_ERR_INVALID_ARGUMENT, "total_reprojection_errors_l1_l2_sharded: communicator and problem live on different devices");
    return errors_l1_l2_impl(p, comm, l1, l2);
    C2B_API_END("problem_total_reprojection_errors_l1_l2_sharded")
}

// Results leave in chunks of kJacChunk observations through a ring of kJacSlots device buffers: the kernel of chunk
// k + 1 is queued before the copies of chunk k start, copies run on their own stream, so PCIe and the kernel overlap
// and the device never holds more than the ring (159 MB) whatever the problem size.  Host buffers from
// c2b_host_alloc (pinned) take the copies at link speed; ordinary pageable memory works too, at the runtime's staged
// rate.  The ring, the copy stream and the events are created on first use and live as long as the problem.
int c2b_problem_residual_jacobian(c2b_problem *p, double *r, double *Jc, double *Jp) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_residual_jacobian");
    if (!p->n_obs) return C2B_OK;
    if (!r || !Jc || !Jp) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_residual_jacobian: NULL output");
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    constexpr int kSlots = c2b_problem::kJacSlots;
    constexpr int64_t kChunk = c2b_problem::kJacChunk;
    if (!p->jac_ring) HIP_TRY(hipMalloc((void **)&p->jac_ring, sizeof(double) * 26 * (size_t)kChunk * kSlots));
    if (!p->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < kSlots; ++k) {
        if (!p->ev_done[k]) HIP_TRY(hipEventCreateWithFlags(&p->ev_done[k], hipEventDisableTiming));
        if (!p->ev_free[k]) HIP_TRY(hipEventCreateWithFlags(&p->ev_free[k], hipEventDisableTiming));
    }
    const int64_t n = p->n_obs, n_chunks = (n + kChunk - 1) / kChunk;
    auto slot_r = [&](int s) { return p->jac_ring + (size_t)s * 26 * kChunk; };
    auto slot_Jc = [&](int s) { return slot_r(s) + 2 * kChunk; };
    auto slot_Jp = [&](int s) { return slot_r(s) + 20 * kChunk; };
    auto launch = [&](int64_t k) -> int {
        const int s = (int)(k % kSlots);
        const int64_t o0 = k * kChunk, m = (n - o0 < kChunk) ? n - o0 : kChunk;
        if (k >= kSlots) { HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_free[s], 0)); }      // its previous copies are out
        // kJacChunk is a multiple of 64: every chunk starts on a tile record
        // (n_pts = 0: a chunk's working set is small and this path is bound by the PCIe copies; loads stay cached)
        int rc2 = c2b_residual_jacobian_rows(p->camblk, p->pts4, 0, p->rows_ptr, p->n_cam, (const char *)p->rows_tiles + (o0 >> 6) * 16, o0,
                                             p->pt_idx + o0, p->uv + 2 * o0, m, slot_r(s), slot_Jc(s), slot_Jp(s), 2.0, nullptr,
                                             nullptr, p->stream);
        if (rc2) return rc2;
        HIP_TRY(hipEventRecord(p->ev_done[s], p->stream));
        return C2B_OK;
    };
    hipError_t e = hipSuccess;
    rc = launch(0);
    for (int64_t k = 0; k < n_chunks && !rc && e == hipSuccess; ++k) {
        if (k + 1 < n_chunks) rc = launch(k + 1);            // queued BEFORE chunk k's copies: they overlap
        if (rc) break;
        const int s = (int)(k % kSlots);
        const int64_t o0 = k * kChunk, m = (n - o0 < kChunk) ? n - o0 : kChunk;
        e = hipStreamWaitEvent(p->copy_stream, p->ev_done[s], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(r + 2 * o0, slot_r(s), sizeof(double) * 2 * m, hipMemcpyDeviceToHost, p->copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(Jc + 18 * o0, slot_Jc(s), sizeof(double) * 18 * m, hipMemcpyDeviceToHost, p->copy_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(Jp + 6 * o0, slot_Jp(s), sizeof(double) * 6 * m, hipMemcpyDeviceToHost, p->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(p->ev_free[s], p->copy_stream);
    }
    const hipError_t e1 = hipStreamSynchronize(p->copy_stream), e2 = hipStreamSynchronize(p->stream);
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2;
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_residual_jacobian: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_residual_jacobian")
}

// The same launch with the results left ON THE DEVICE, in output arrays placed for streaming stores: what a
// BAProblem-level caller that consumes the Jacobian on the GPU (a solver's normal equations) calls in its loop.  The
// whole list in ONE launch -- residual, both blocks and the folded sum of squared residuals -- at the Level-0 headline
// rate; nothing crosses PCIe but the 8-byte sum.  *outputs == NULL: a set is allocated by c2b_jacobian_outputs_alloc
// (max_attempts placements tried, as there) and handed to the caller, who passes it back on later calls (it is reused as
// long as the observation count matches) and frees it with c2b_jacobian_outputs_free.
int c2b_problem_residual_jacobian_device(c2b_problem *p, int max_attempts, c2b_jacobian_outputs **outputs, double *sum_sq) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_residual_jacobian_device");
    if (!outputs) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_residual_jacobian_device: outputs is NULL");
    if (*outputs && ((*outputs)->n_obs != p->n_obs || (*outputs)->device != p->device))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_residual_jacobian_device: the output set holds %lld observations on device %d, the problem %lld on device %d",
                    (long long)(*outputs)->n_obs, (*outputs)->device, (long long)p->n_obs, p->device);
    int rc = ensure_camblk(p);
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    const bool mine = *outputs == nullptr;
    if (mine) {
        rc = c2b_jacobian_outputs_alloc(p->n_obs, max_attempts, 0.0, p->stream, outputs);
        if (rc) return rc;
    }
    c2b_jacobian_outputs *h = *outputs;
    rc = c2b_residual_jacobian_rows(p->camblk, p->pts4, p->n_pts, p->rows_ptr, p->n_cam, p->rows_tiles, 0, p->pt_idx, p->uv, p->n_obs,
                                    h->r, h->Jc, h->Jp, 2.0, p->ws, p->scalar, p->stream);
    double sum = 0.0;
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(&sum, p->scalar, sizeof(double), hipMemcpyDeviceToHost, p->stream);
    const hipError_t e2 = hipStreamSynchronize(p->stream);
    if (rc || e != hipSuccess || e2 != hipSuccess) {
        if (mine) { c2b_jacobian_outputs_free(h); *outputs = nullptr; }
        if (rc) return rc;
        return fail(C2B_ERR_HIP, "problem_residual_jacobian_device: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    if (sum_sq) *sum_sq = sum;
    return C2B_OK;
    C2B_API_END("problem_residual_jacobian_device")
}

int c2b_host_alloc(void **ptr, int64_t bytes) {
    C2B_API_BEGIN
    if (!ptr || bytes < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "host_alloc: bad arguments");
    *ptr = nullptr;
    if (!bytes) return C2B_OK;
    HIP_TRY(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return C2B_OK;
    C2B_API_END("host_alloc")
}

void c2b_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

static int compute_stats(c2b_problem *p) {
    int rc = ensure_camblk(p);
    if (rc) return rc;
    return c2b_stats(p->camblk, p->n_cam, p->pts4, p->n_pts, p->ws, p->stats, p->stream);
}

int c2b_problem_stats(c2b_problem *p, double *stats) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_stats");
    if (!stats) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_stats: stats is NULL");
    int rc = compute_stats(p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stats, p->stats, sizeof(double) * C2B_STATS_DOUBLES, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_stats")
}

int c2b_problem_visibility_pairs(c2b_problem *p, int64_t n_pairs, const uint32_t *cam_idx, const uint32_t *pt_idx,
                                 double max_dist, double *uv_out, uint8_t *keep) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_pairs");
    if (n_pairs < 0 || (n_pairs && (!cam_idx || !pt_idx || !uv_out || !keep)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_pairs: bad arguments");
    if (!n_pairs) return C2B_OK;
    for (int64_t i = 0; i < n_pairs; ++i)
        if (cam_idx[i] >= (uint64_t)p->n_cam || pt_idx[i] >= (uint64_t)p->n_pts)
            return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "problem_visibility_pairs: pair %lld out of range", (long long)i);
    int rc = ensure_camblk(p);
    if (rc) return rc;
    uint32_t *d_c = nullptr, *d_p = nullptr;
    double *d_uv = nullptr;
    uint8_t *d_k = nullptr;
    hipError_t e = hipMalloc((void **)&d_c, sizeof(uint32_t) * n_pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&d_p, sizeof(uint32_t) * n_pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&d_uv, sizeof(double) * 2 * n_pairs);
    if (e == hipSuccess) e = hipMalloc((void **)&d_k, n_pairs);
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, cam_idx, sizeof(uint32_t) * n_pairs, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_p, pt_idx, sizeof(uint32_t) * n_pairs, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        rc = c2b_visibility_pairs(p->camblk, p->pts4, d_c, d_p, n_pairs, max_dist, d_uv, d_k, p->stream);
        if (!rc) {
            e = hipMemcpyAsync(uv_out, d_uv, sizeof(double) * 2 * n_pairs, hipMemcpyDeviceToHost, p->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(keep, d_k, n_pairs, hipMemcpyDeviceToHost, p->stream);
        }
        hipError_t e2 = hipStreamSynchronize(p->stream);
        if (e == hipSuccess) e = e2;
    }
    if (d_c) (void)hipFree(d_c);
    if (d_p) (void)hipFree(d_p);
    if (d_uv) (void)hipFree(d_uv);
    if (d_k) (void)hipFree(d_k);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_pairs: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_visibility_pairs")
}

/* ---- BAProblem::cull on the device (src/baproblem.rs:538-549) ---- */
extern "C++" {
namespace {

// device allocation that frees itself (the cull pipeline holds ~20 scratch arrays)
struct DevBuf {
    void *ptr = nullptr;
    bool owned = true;                         // false: a view into an arena (below), never freed or released by itself
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (ptr && owned) (void)hipFree(ptr); }
    hipError_t alloc(size_t bytes) { owned = true; return hipMalloc(&ptr, bytes ? bytes : 16); }
    void view(void *q) { ptr = q; owned = false; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(ptr); }
    void *release() { void *q = ptr; ptr = nullptr; return q; }
};

// Temporaries of one call carved out of ONE allocation: a device malloc / free pair costs ~1 ms at these sizes (the free
// synchronises), and cull used to make ~25 of each -- most of its 45 ms at --blocks 128 once its kernels took 10.
struct DevArena {
    DevBuf block;
    size_t used = 0, cap = 0;
    static size_t rounded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    hipError_t reserve(size_t bytes) { cap = bytes; return block.alloc(bytes); }
    void *take(size_t bytes) {
        void *q = static_cast<char *>(block.ptr) + used;
        used += rounded(bytes ? bytes : 16);
        return used <= cap ? q : nullptr;
    }
};

unsigned blocks_of(int64_t n, int per) { return (unsigned)((n + per - 1) / per > 0 ? (n + per - 1) / per : 1); }

// exclusive scan of n 0/1 flags into pos; *total_host = number of set flags.  Synchronises.
hipError_t scan_flags(hipStream_t st, const uint32_t *flags, int64_t n, uint32_t *pos, uint32_t *tile_scratch, uint32_t *d_total,
                      uint32_t *total_host) {
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    if (n > 0) hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, flags, n, pos, tile_scratch);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kScanBlock), 0, st, tile_scratch, tiles, d_total);
    if (n > 0) hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, pos, n, (const uint32_t *)tile_scratch);
    hipError_t e = launch_error();
    if (e == hipSuccess) e = hipMemcpyAsync(total_host, d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e;
}

}  // namespace
}  // extern "C++"

// mode 0: cull() = both passes to a fixed point; 1: largest_connected_component() once; 2: remove_singletons() once
static int cull_impl(c2b_problem *p, int faithful, int mode) {
    NEED_UPLOADED(p, "problem_cull");
    if (p->n_obs >= ((int64_t)1 << 32) || p->n_cam + p->n_pts >= ((int64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_cull: more than 2^32 observations or entities");
    free_dense(p);
    hipStream_t st = p->stream;
    const int64_t nc0 = p->n_cam, np0 = p->n_pts, no0 = p->n_obs;
    const int64_t nodes0 = nc0 + np0, big0 = std::max(std::max(nc0, np0), no0);
    // current graph (ping-pong pairs) + where everything came from
    DevBuf cam[2], pt[2], eorig[2], corig[2], porig[2];
    DevBuf parent, sets, size, keep_c, keep_p, keep_o, pos_c, pos_p, pos_o, tiles, best, total, deg, cnt;
    hipError_t e = hipSuccess;
    auto A = [&](DevBuf &b, size_t bytes) { if (e == hipSuccess) e = b.alloc(bytes); };
    // cam / pt are allocations of their own (one of each pair becomes the problem's index array); every other temporary
    // is a view into one arena
    for (int k = 0; k < 2; ++k) { A(cam[k], 4 * (size_t)no0); A(pt[k], 4 * (size_t)no0); }
    DevArena arena;
    struct Want { DevBuf *b; size_t bytes; };
    const Want wants[] = {
        {&eorig[0], 4 * (size_t)no0}, {&eorig[1], 4 * (size_t)no0}, {&corig[0], 4 * (size_t)nc0}, {&corig[1], 4 * (size_t)nc0},
        {&porig[0], 4 * (size_t)np0}, {&porig[1], 4 * (size_t)np0},
        {&parent, 4 * (size_t)nodes0}, {&sets, 4 * (size_t)nodes0}, {&size, 4 * (size_t)nodes0},
        {&keep_c, 4 * (size_t)nc0}, {&keep_p, 4 * (size_t)np0}, {&keep_o, 4 * (size_t)no0},
        {&pos_c, 4 * (size_t)nc0}, {&pos_p, 4 * (size_t)np0}, {&pos_o, 4 * (size_t)no0},
        {&tiles, 4 * (size_t)(big0 / kScanTile + 2)}, {&best, 8}, {&total, 4}, {&deg, 4 * (size_t)nc0}, {&cnt, 4 * (size_t)np0}};
    size_t arena_bytes = 0;
    for (const Want &w : wants) arena_bytes += DevArena::rounded(w.bytes ? w.bytes : 16);
    if (e == hipSuccess) e = arena.reserve(arena_bytes);
    if (e == hipSuccess)
        for (const Want &w : wants) w.b->view(arena.take(w.bytes));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_cull: %s", hipGetErrorString(e));

    int cur = 0;
    int64_t nc = nc0, np = np0, no = no0;
    if (no) {
        e = hipMemcpyAsync(cam[0].ptr, p->cam_idx, 4 * (size_t)no, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(pt[0].ptr, p->pt_idx, 4 * (size_t)no, hipMemcpyDeviceToDevice, st);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_uf_init, dim3(blocks_of(no, kBlock)), dim3(kBlock), 0, st, eorig[0].as<uint32_t>(), no);   // iota
        hipLaunchKernelGGL(k_uf_init, dim3(blocks_of(nc, kBlock)), dim3(kBlock), 0, st, corig[0].as<uint32_t>(), nc);
        hipLaunchKernelGGL(k_uf_init, dim3(blocks_of(np, kBlock)), dim3(kBlock), 0, st, porig[0].as<uint32_t>(), np);
        e = launch_error();
    }

    // renumber by the keep flags currently in keep_c / keep_p / keep_o
    auto compact = [&]() -> hipError_t {
        uint32_t nc_new = 0, np_new = 0, no_new = 0;
        hipError_t s = scan_flags(st, keep_c.as<uint32_t>(), nc, pos_c.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &nc_new);
        if (s == hipSuccess) s = scan_flags(st, keep_p.as<uint32_t>(), np, pos_p.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &np_new);
        if (s == hipSuccess) s = scan_flags(st, keep_o.as<uint32_t>(), no, pos_o.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &no_new);
        if (s != hipSuccess) return s;
        const int nxt = cur ^ 1;
        hipLaunchKernelGGL(k_cull_move_nodes, dim3(blocks_of(nc, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)keep_c.as<uint32_t>(),
                           (const uint32_t *)pos_c.as<uint32_t>(), nc, (const uint32_t *)corig[cur].as<uint32_t>(), corig[nxt].as<uint32_t>());
        hipLaunchKernelGGL(k_cull_move_nodes, dim3(blocks_of(np, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)keep_p.as<uint32_t>(),
                           (const uint32_t *)pos_p.as<uint32_t>(), np, (const uint32_t *)porig[cur].as<uint32_t>(), porig[nxt].as<uint32_t>());
        hipLaunchKernelGGL(k_cull_move_edges, dim3(blocks_of(no, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)keep_o.as<uint32_t>(),
                           (const uint32_t *)pos_o.as<uint32_t>(), no, (const uint32_t *)cam[cur].as<uint32_t>(),
                           (const uint32_t *)pt[cur].as<uint32_t>(), (const uint32_t *)eorig[cur].as<uint32_t>(),
                           (const uint32_t *)pos_c.as<uint32_t>(), (const uint32_t *)pos_p.as<uint32_t>(), cam[nxt].as<uint32_t>(),
                           pt[nxt].as<uint32_t>(), eorig[nxt].as<uint32_t>());
        cur = nxt;
        nc = nc_new; np = np_new; no = no_new;
        return launch_error();
    };
    auto lcc_pass = [&]() -> hipError_t {
        if (nc == 0) return hipSuccess;                      // largest_connected_component returns self (:457-459)
        const int64_t nodes = nc + np, big = std::max(std::max(nc, np), no);
        hipError_t s = hipMemsetAsync(size.ptr, 0, 4 * (size_t)nodes, st);
        if (s == hipSuccess) s = hipMemsetAsync(best.ptr, 0, 8, st);
        if (s != hipSuccess) return s;
        hipLaunchKernelGGL(k_uf_init, dim3(blocks_of(nodes, kBlock)), dim3(kBlock), 0, st, parent.as<uint32_t>(), nodes);
        if (no) hipLaunchKernelGGL(k_uf_union, dim3(blocks_of(no, kBlock)), dim3(kBlock), 0, st, parent.as<uint32_t>(),
                                   (const uint32_t *)cam[cur].as<uint32_t>(), (const uint32_t *)pt[cur].as<uint32_t>(), no, (uint32_t)nc);
        hipLaunchKernelGGL(k_uf_flatten, dim3(blocks_of(nodes, kBlock)), dim3(kBlock), 0, st, parent.as<uint32_t>(), nodes,
                           sets.as<uint32_t>(), size.as<uint32_t>());
        hipLaunchKernelGGL(k_uf_largest, dim3(blocks_of(nodes, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)sets.as<uint32_t>(),
                           (const uint32_t *)size.as<uint32_t>(), nodes, best.as<unsigned long long>());
        hipLaunchKernelGGL(k_lcc_flags, dim3(blocks_of(big, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)sets.as<uint32_t>(),
                           (const unsigned long long *)best.as<unsigned long long>(), (uint32_t)nc, (uint32_t)np,
                           (const uint32_t *)cam[cur].as<uint32_t>(), (const uint32_t *)pt[cur].as<uint32_t>(), no, faithful ? 1 : 0,
                           keep_c.as<uint32_t>(), keep_p.as<uint32_t>(), keep_o.as<uint32_t>());
        s = launch_error();
        return s == hipSuccess ? compact() : s;
    };
    auto singleton_pass = [&]() -> hipError_t {
        const int64_t big = std::max(std::max(nc, np), no);
        hipError_t s = hipMemsetAsync(deg.ptr, 0, 4 * (size_t)(nc ? nc : 1), st);
        if (s == hipSuccess) s = hipMemsetAsync(cnt.ptr, 0, 4 * (size_t)(np ? np : 1), st);
        if (s != hipSuccess) return s;
        if (no) hipLaunchKernelGGL(k_degree, dim3(blocks_of(no, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)cam[cur].as<uint32_t>(),
                                   (const uint32_t *)pt[cur].as<uint32_t>(), no, deg.as<uint32_t>(), cnt.as<uint32_t>());
        hipLaunchKernelGGL(k_singleton_flags, dim3(blocks_of(big, kBlock)), dim3(kBlock), 0, st, (const uint32_t *)deg.as<uint32_t>(),
                           (const uint32_t *)cnt.as<uint32_t>(), (uint32_t)nc, (uint32_t)np, (const uint32_t *)cam[cur].as<uint32_t>(),
                           (const uint32_t *)pt[cur].as<uint32_t>(), no, keep_c.as<uint32_t>(), keep_p.as<uint32_t>(), keep_o.as<uint32_t>());
        s = launch_error();
        return s == hipSuccess ? compact() : s;
    };
    // culled = lcc().remove_singletons(); while the counts change: again (src/baproblem.rs:541-547)
    int64_t pnc = nc, pnp = np;
    if (e == hipSuccess && mode != 2) e = lcc_pass();
    if (e == hipSuccess && mode != 1) e = singleton_pass();
    while (mode == 0 && e == hipSuccess && (nc != pnc || np != pnp)) {
        pnc = nc; pnp = np;
        e = lcc_pass();
        if (e == hipSuccess) e = singleton_pass();
    }

    // gather the payloads once and swap them in
    DevBuf n_cam15, n_bal9, n_camblk, n_pts4, n_uv, n_ws;
    if (e == hipSuccess) {
        A(n_cam15, sizeof(double) * 15 * (size_t)nc); A(n_bal9, sizeof(double) * 9 * (size_t)nc);
        A(n_camblk, sizeof(double) * kCamBlk * (size_t)nc); A(n_pts4, sizeof(double) * 4 * (size_t)np);
        A(n_uv, sizeof(double) * 2 * (size_t)no); A(n_ws, (size_t)c2b_workspace_bytes(no));
    }
    if (e == hipSuccess && c2b_workspace_init(n_ws.ptr, st) != C2B_OK) e = hipErrorUnknown;
    if (e == hipSuccess) {
        auto gather = [&](const double *in, const DevBuf &orig, int64_t n, int width, DevBuf &out) {
            if (n) hipLaunchKernelGGL(k_gather_rows, dim3(blocks_of(n * width, kBlock)), dim3(kBlock), 0, st, in,
                                      (const uint32_t *)orig.as<uint32_t>(), n, width, out.as<double>());
        };
        gather(p->cam15, corig[cur], nc, 15, n_cam15);
        if (p->bal_valid) gather(p->bal9, corig[cur], nc, 9, n_bal9);
        gather(p->pts4, porig[cur], np, 4, n_pts4);
        gather(p->uv, eorig[cur], no, 2, n_uv);
        e = launch_error();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_cull: %s", hipGetErrorString(e));
    }
    void *old[] = {p->cam15, p->bal9, p->camblk, p->pts4, p->uv, p->cam_idx, p->pt_idx, p->ws};
    for (void *q : old) if (q) (void)hipFree(q);
    p->cam15 = (double *)n_cam15.release(); p->bal9 = (double *)n_bal9.release(); p->camblk = (double *)n_camblk.release();
    p->pts4 = (double *)n_pts4.release(); p->uv = (double *)n_uv.release(); p->ws = n_ws.release();
    p->cam_idx = (uint32_t *)cam[cur].release(); p->pt_idx = (uint32_t *)pt[cur].release();
    drop_rows(p);
    p->n_cam = nc; p->n_pts = np; p->n_obs = no;
    p->blk_valid = false;                                     // camblk is rebuilt on demand from the gathered cameras
    return C2B_OK;
}

int c2b_problem_cull(c2b_problem *p, int faithful) { return cull_impl(p, faithful, 0); }
int c2b_problem_largest_connected_component(c2b_problem *p, int faithful) { return cull_impl(p, faithful, 1); }
int c2b_problem_remove_singletons(c2b_problem *p) { return cull_impl(p, 1, 2); }

int c2b_problem_adopt_visibility(c2b_problem *p) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_adopt_visibility");
    if (!p->dense_pt || !p->dense_uv || !p->dense_row)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_adopt_visibility: no pending visibility result");
    const int64_t n = p->dense_n;
    if (n >= ((int64_t)1 << 32)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_adopt_visibility: more than 2^32 observations");
    DevBuf cam_idx, ws;
    hipError_t e = cam_idx.alloc(sizeof(uint32_t) * (size_t)n);
    if (e == hipSuccess) e = ws.alloc((size_t)c2b_workspace_bytes(n));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_adopt_visibility: %s", hipGetErrorString(e));
    int rc = c2b_expand_rows(p->dense_row, p->n_cam, 0, n, cam_idx.as<uint32_t>(), p->stream);
    if (!rc) rc = c2b_workspace_init(ws.ptr, p->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    void *old[] = {p->uv, p->cam_idx, p->pt_idx, p->ws, p->dense_row};
    for (void *q : old) if (q) (void)hipFree(q);
    p->cam_idx = (uint32_t *)cam_idx.release();
    drop_rows(p);
    p->ws = ws.release();
    p->pt_idx = p->dense_pt; p->uv = p->dense_uv; p->n_obs = n;
    p->dense_pt = nullptr; p->dense_uv = nullptr; p->dense_row = nullptr; p->dense_n = 0;
    return C2B_OK;
    C2B_API_END("problem_adopt_visibility")
}

int c2b_problem_download_graph(c2b_problem *p, uint64_t *row_ptr, uint64_t *pt_idx) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_download_graph");
    if (!row_ptr || (p->n_obs && !pt_idx)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_download_graph: bad arguments");
    const int64_t n_cam = p->n_cam, n_obs = p->n_obs;
    // the point indices are widened to the host's u64 on the device and leave in ONE copy (r01-r02: a u32 copy into a
    // fresh host vector, then a serial widening loop over 19 M entries -- a third of the 120-ms download at --blocks 128)
    DevBuf d_row, d_pt64;
    hipError_t e = d_row.alloc(sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess && n_obs) e = d_pt64.alloc(sizeof(uint64_t) * (size_t)n_obs);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rows_from_sorted, dim3(blocks_for(n_obs + 1)), dim3(kBlock), 0, p->stream, (const uint32_t *)p->cam_idx, n_obs,
                           n_cam, d_row.as<uint64_t>());
        if (n_obs) hipLaunchKernelGGL(k_widen_u32, dim3(blocks_for(n_obs)), dim3(kBlock), 0, p->stream, (const uint32_t *)p->pt_idx, n_obs,
                                      d_pt64.as<uint64_t>());
        e = launch_error();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(row_ptr, d_row.ptr, sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess && n_obs) e = hipMemcpyAsync(pt_idx, d_pt64.ptr, sizeof(uint64_t) * (size_t)n_obs, hipMemcpyDeviceToHost, p->stream);
    hipError_t e2 = hipStreamSynchronize(p->stream);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_download_graph: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_download_graph")
}

// Stable compaction of CSR lists (point index, uv) by a keep mask, on the device: kept count per row -> row scan ->
// one wave per camera scatters in order.  The new row pointer goes to row_ptr_host; on success the three new device
// buffers are handed to the caller (who owns them), *w = kept count.  Synchronises the problem's stream.
static hipError_t compact_rows_on_device(c2b_problem *p, const uint64_t *d_row_old, const uint8_t *d_keep, const uint32_t *d_pt,
                                         const double *d_uv, int64_t n_cam, uint64_t *row_ptr_host, uint64_t **d_row_new,
                                         uint32_t **d_pt_new, double **d_uv_new, int64_t *w) {
    uint64_t *d_tot = nullptr;
    *d_row_new = nullptr; *d_pt_new = nullptr; *d_uv_new = nullptr; *w = 0;
    hipError_t e = hipMalloc((void **)&d_tot, sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess) e = hipMalloc((void **)d_row_new, sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess) {
        const unsigned row_blocks = (unsigned)((n_cam + 3) / 4);
        if (n_cam) hipLaunchKernelGGL(k_keep_row_counts, dim3(row_blocks), dim3(256), 0, p->stream, d_row_old, d_keep, n_cam, d_tot);
        hipLaunchKernelGGL(k_dense_cam_scan, dim3(1), dim3(256), 0, p->stream, (const uint64_t *)d_tot, n_cam, *d_row_new);
        e = hipMemcpyAsync(row_ptr_host, *d_row_new, sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyDeviceToHost, p->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        if (e == hipSuccess) {
            *w = (int64_t)row_ptr_host[n_cam];
            e = hipMalloc((void **)d_pt_new, sizeof(uint32_t) * (size_t)(*w ? *w : 4));
            if (e == hipSuccess) e = hipMalloc((void **)d_uv_new, sizeof(double) * 2 * (size_t)(*w ? *w : 1));
            if (e == hipSuccess && n_cam) {
                hipLaunchKernelGGL(k_keep_row_scatter, dim3(row_blocks), dim3(256), 0, p->stream, d_row_old,
                                   (const uint64_t *)*d_row_new, d_keep, d_pt, reinterpret_cast<const double2 *>(d_uv), n_cam,
                                   *d_pt_new, reinterpret_cast<double2 *>(*d_uv_new));
                e = launch_error();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        }
    }
    if (d_tot) (void)hipFree(d_tot);
    if (e != hipSuccess) {
        if (*d_row_new) (void)hipFree(*d_row_new);
        if (*d_pt_new) (void)hipFree(*d_pt_new);
        if (*d_uv_new) (void)hipFree(*d_uv_new);
        *d_row_new = nullptr; *d_pt_new = nullptr; *d_uv_new = nullptr;
    }
    return e;
}

int c2b_problem_visibility_pairs_compact(c2b_problem *p, int64_t n_pairs, const uint32_t *cam_idx, const uint32_t *pt_idx,
                                         double max_dist, uint64_t *row_ptr) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_pairs_compact");
    if (n_pairs < 0 || !row_ptr || (n_pairs && (!cam_idx || !pt_idx)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_pairs_compact: bad arguments");
    for (int64_t i = 0; i < n_pairs; ++i) {
        if (cam_idx[i] >= (uint64_t)p->n_cam || pt_idx[i] >= (uint64_t)p->n_pts)
            return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "problem_visibility_pairs_compact: pair %lld out of range", (long long)i);
        if (i && cam_idx[i] < cam_idx[i - 1])
            return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_pairs_compact: cam_idx must be non-decreasing (pair %lld)", (long long)i);
    }
    int rc = ensure_camblk(p);
    if (rc) return rc;
    free_dense(p);
    const int64_t n_cam = p->n_cam;
    const size_t n = (size_t)(n_pairs ? n_pairs : 1);
    uint32_t *d_c = nullptr, *d_p = nullptr, *d_pt_new = nullptr;
    double *d_uv = nullptr, *d_uv_new = nullptr;
    uint8_t *d_k = nullptr;
    uint64_t *d_row = nullptr, *d_row_new = nullptr;
    int64_t w = 0;
    hipError_t e = hipMalloc((void **)&d_c, sizeof(uint32_t) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_p, sizeof(uint32_t) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_uv, sizeof(double) * 2 * n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_k, n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_row, sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess && n_pairs) e = hipMemcpyAsync(d_c, cam_idx, sizeof(uint32_t) * n, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess && n_pairs) e = hipMemcpyAsync(d_p, pt_idx, sizeof(uint32_t) * n, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        if (n_pairs) rc = c2b_visibility_pairs(p->camblk, p->pts4, d_c, d_p, n_pairs, max_dist, d_uv, d_k, p->stream);
        if (!rc) {
            hipLaunchKernelGGL(k_rows_from_sorted, dim3(blocks_for(n_pairs + 1)), dim3(kBlock), 0, p->stream, (const uint32_t *)d_c,
                               n_pairs, n_cam, d_row);
            e = launch_error();
            if (e == hipSuccess)
                e = compact_rows_on_device(p, d_row, d_k, d_p, d_uv, n_cam, row_ptr, &d_row_new, &d_pt_new, &d_uv_new, &w);
        }
    }
    if (rc || e != hipSuccess) (void)hipStreamSynchronize(p->stream);
    if (!rc && e == hipSuccess) {                 // becomes the pending visibility result (fetch with _dense_fetch)
        p->dense_row = d_row_new; p->dense_pt = d_pt_new; p->dense_uv = d_uv_new; p->dense_n = w;
    }
    if (d_c) (void)hipFree(d_c);
    if (d_p) (void)hipFree(d_p);
    if (d_uv) (void)hipFree(d_uv);
    if (d_k) (void)hipFree(d_k);
    if (d_row) (void)hipFree(d_row);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_pairs_compact: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_visibility_pairs_compact")
}

// The generators' whole visibility loop (src/synthetic.rs:268-297, :353-378) on the device: candidates by a cell list
// (rstar's locate_within_distance), the sight line against the buildings (hits_building), the predicate, and the kept
// (point, uv) lists compacted per camera in ascending point index -- csrc/cell_kernels.hpp.  The result becomes the
// pending visibility result like c2b_problem_visibility_pairs_compact's (adopt / fetch it the same way); row_ptr (host,
// n_cam + 1) may be NULL.
int c2b_problem_visibility_within_distance(c2b_problem *p, double max_dist, int occlusion, double block_length, double block_inset,
                                           uint64_t *row_ptr) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_within_distance");
    if (!(max_dist >= 0.0) || (occlusion && !(block_length > 0.0)))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_within_distance: max_dist must be >= 0 (and block_length > 0 with occlusion)");
    const int64_t n_cam = p->n_cam, n_pts = p->n_pts;
    if (n_pts >= ((int64_t)1 << 32) || n_cam >= ((int64_t)1 << 31))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_within_distance: too many cameras or points");
    int rc = ensure_camblk(p);
    if (rc) return rc;
    free_dense(p);
    hipStream_t st = p->stream;
    if (n_cam == 0 || n_pts == 0) {                                      // nothing can be seen: an empty graph
        DevBuf row0, pt0, uv0;
        hipError_t e0 = row0.alloc(sizeof(uint64_t) * (size_t)(n_cam + 1));
        if (e0 == hipSuccess) e0 = pt0.alloc(4);
        if (e0 == hipSuccess) e0 = uv0.alloc(16);
        if (e0 == hipSuccess) e0 = hipMemsetAsync(row0.ptr, 0, sizeof(uint64_t) * (size_t)(n_cam + 1), st);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(st);
        if (e0 != hipSuccess) return fail(C2B_ERR_HIP, "problem_visibility_within_distance: %s", hipGetErrorString(e0));
        if (row_ptr) std::fill(row_ptr, row_ptr + n_cam + 1, (uint64_t)0);
        p->dense_row = (uint64_t *)row0.release(); p->dense_pt = (uint32_t *)pt0.release(); p->dense_uv = (double *)uv0.release();
        p->dense_n = 0;
        return C2B_OK;
    }
    // extent of cameras and points -> the cell grid.  Cells are a hair wider than max_dist so that rounding in the cell
    // arithmetic can never separate a camera from a point within max_dist by more than one cell.
    double stats[C2B_STATS_DOUBLES];
    rc = compute_stats(p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stats, p->stats, sizeof stats, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    CellGrid g;
    g.x0 = stats[6]; g.z0 = stats[8];
    const double ex = stats[9] - stats[6], ez = stats[11] - stats[8];
    double cs = (max_dist > 0.0 ? max_dist : 1.0) * (1.0 + 0x1.0p-20);
    if (!(ex >= 0.0) || !(ez >= 0.0) || !std::isfinite(ex) || !std::isfinite(ez) || !std::isfinite(cs))
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_within_distance: non-finite coordinates");
    auto cells = [&](double c) { return (std::floor(ex / c) + 1.0) * (std::floor(ez / c) + 1.0); };
    while (cells(cs) > (double)(1 << 24)) cs *= 2.0;                     // wider cells stay correct, only slower
    g.inv_cs = 1.0 / cs;
    g.ncx = (int)std::floor(ex / cs) + 1; g.ncz = (int)std::floor(ez / cs) + 1;
    const int64_t n_cells = (int64_t)g.ncx * g.ncz;

    DevArena arena;
    DevBuf cell_of, counts, cursor, sorted, tiles, total, cam_count, pos, sum64, row64;
    struct Want { DevBuf *b; size_t bytes; };
    const int64_t big = std::max(n_cells + 1, n_cam + 1);
    const Want wants[] = {{&cell_of, 4 * (size_t)n_pts}, {&counts, 4 * (size_t)(n_cells + 1)}, {&cursor, 4 * (size_t)(n_cells + 1)},
                          {&sorted, 4 * (size_t)n_pts}, {&tiles, 4 * (size_t)(big / kScanTile + 2)}, {&total, 4},
                          {&cam_count, 4 * (size_t)(n_cam + 1)}, {&pos, 4 * (size_t)(n_cam + 1)}, {&sum64, 8}};
    size_t arena_bytes = 0;
    for (const Want &w : wants) arena_bytes += DevArena::rounded(w.bytes ? w.bytes : 16);
    hipError_t e = arena.reserve(arena_bytes);
    if (e == hipSuccess) e = row64.alloc(sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_within_distance: %s", hipGetErrorString(e));
    for (const Want &w : wants) w.b->view(arena.take(w.bytes));

    // cell list: count, exclusive scan (start[n_cells] = n_pts), fill
    HIP_TRY(hipMemsetAsync(counts.ptr, 0, 4 * (size_t)(n_cells + 1), st));
    HIP_TRY(hipMemsetAsync(cursor.ptr, 0, 4 * (size_t)(n_cells + 1), st));
    HIP_TRY(hipMemsetAsync(cam_count.ptr, 0, 4 * (size_t)(n_cam + 1), st));
    HIP_TRY(hipMemsetAsync(sum64.ptr, 0, 8, st));
    if (n_pts) hipLaunchKernelGGL(k_cells_assign, dim3(blocks_of(n_pts, 256)), dim3(256), 0, st, reinterpret_cast<const double4 *>(p->pts4), n_pts,
                                  g, cell_of.as<uint32_t>(), counts.as<uint32_t>());
    uint32_t n_sorted = 0, n_kept32 = 0;
    uint32_t *start = cursor.as<uint32_t>();                             // scanned counts; the fill's cursors live in `counts` afterwards
    e = scan_flags(st, counts.as<uint32_t>(), n_cells + 1, start, tiles.as<uint32_t>(), total.as<uint32_t>(), &n_sorted);
    if (e == hipSuccess && (int64_t)n_sorted != n_pts) return fail(C2B_ERR_HIP, "problem_visibility_within_distance: cell counts do not add up");
    if (e == hipSuccess) e = hipMemsetAsync(counts.ptr, 0, 4 * (size_t)(n_cells + 1), st);
    if (e == hipSuccess && n_pts)
        hipLaunchKernelGGL(k_cells_fill, dim3(blocks_of(n_pts, 256)), dim3(256), 0, st, (const uint32_t *)cell_of.as<uint32_t>(), n_pts,
                           (const uint32_t *)start, counts.as<uint32_t>(), sorted.as<uint32_t>());
    // pass 1: survivors per camera; scan; total
    const unsigned cam_blocks = blocks_of(n_cam, kCellWPB);
    if (e == hipSuccess && n_cam && n_pts) {
        hipLaunchKernelGGL((k_cells_visibility<false>), dim3(cam_blocks), dim3(kCellWPB * 64), 0, st, (const double *)p->camblk, n_cam,
                           reinterpret_cast<const double4 *>(p->pts4), g, (const uint32_t *)start, (const uint32_t *)sorted.as<uint32_t>(),
                           max_dist, occlusion ? 1 : 0, block_length, block_inset, cam_count.as<uint32_t>(), (const uint64_t *)nullptr,
                           (uint32_t *)nullptr, (double2 *)nullptr);
        hipLaunchKernelGGL(k_sum_u32_u64, dim3(256), dim3(256), 0, st, (const uint32_t *)cam_count.as<uint32_t>(), n_cam,
                           sum64.as<unsigned long long>());
    }
    if (e == hipSuccess) e = scan_flags(st, cam_count.as<uint32_t>(), n_cam + 1, pos.as<uint32_t>(), tiles.as<uint32_t>(), total.as<uint32_t>(), &n_kept32);
    unsigned long long n_kept = 0;
    if (e == hipSuccess) e = hipMemcpy(&n_kept, sum64.ptr, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && n_kept != (unsigned long long)n_kept32)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_within_distance: more than 2^32 observations");
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_widen_u32, dim3(blocks_for(n_cam + 1)), dim3(kBlock), 0, st, (const uint32_t *)pos.as<uint32_t>(), n_cam + 1,
                           row64.as<uint64_t>());
        e = launch_error();
    }
    // pass 2: fill in meeting order, then every row into ascending point index
    DevBuf tmp_pt, tmp_uv, out_pt, out_uv;
    const size_t w = (size_t)n_kept;
    if (e == hipSuccess) e = tmp_pt.alloc(4 * w);
    if (e == hipSuccess) e = tmp_uv.alloc(16 * w);
    if (e == hipSuccess) e = out_pt.alloc(4 * w);
    if (e == hipSuccess) e = out_uv.alloc(16 * w);
    if (e == hipSuccess && w) {
        hipLaunchKernelGGL((k_cells_visibility<true>), dim3(cam_blocks), dim3(kCellWPB * 64), 0, st, (const double *)p->camblk, n_cam,
                           reinterpret_cast<const double4 *>(p->pts4), g, (const uint32_t *)start, (const uint32_t *)sorted.as<uint32_t>(),
                           max_dist, occlusion ? 1 : 0, block_length, block_inset, (uint32_t *)nullptr, (const uint64_t *)row64.as<uint64_t>(),
                           tmp_pt.as<uint32_t>(), tmp_uv.as<double2>());
        hipLaunchKernelGGL(k_rows_rank_sort, dim3(cam_blocks), dim3(kCellWPB * 64), 0, st, (const uint64_t *)row64.as<uint64_t>(), n_cam,
                           (const uint32_t *)tmp_pt.as<uint32_t>(), (const double2 *)tmp_uv.as<double2>(), out_pt.as<uint32_t>(),
                           out_uv.as<double2>());
        e = launch_error();
    }
    if (e == hipSuccess && row_ptr)
        e = hipMemcpyAsync(row_ptr, row64.ptr, sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_within_distance: %s", hipGetErrorString(e));
    }
    p->dense_row = (uint64_t *)row64.release();
    p->dense_pt = (uint32_t *)out_pt.release();
    p->dense_uv = (double *)out_uv.release();
    p->dense_n = (int64_t)n_kept;
    return C2B_OK;
    C2B_API_END("problem_visibility_within_distance")
}

// BAProblem::write (src/baproblem.rs:768-785) of the RESIDENT problem.  `.bbal` (format 1): the file image is assembled
// on the device (cell_kernels.hpp: k_bbal_*: to_vec of every camera, the per-camera counts, the byte order) and leaves
// through a few host threads, each copying its chunks into a pinned buffer and pwrite()-ing them -- the host touches no
// observation.  `.bal` (format 0): the text writer of csrc/host_baproblem.hpp over a download (shortest round-trip
// decimals are host work).  format -1: by extension, like the reference.
int c2b_problem_write(c2b_problem *p, const char *path, int format) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_write");
    if (!path) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_write: path is NULL");
    bool binary = false;
    int rc = bal_format(path, format, &binary);
    if (rc) return rc;
    const int64_t n_cam = p->n_cam, n_pts = p->n_pts, n_obs = p->n_obs;
    if (!p->bal_valid) {                                     // to_vec (src/baproblem.rs:189-202) of the current state
        rc = c2b_cameras_to_bal(p->cam15, n_cam, p->bal9, p->stream);
        if (rc) return rc;
    }
    if (!binary) {
        std::vector<double> bal9((size_t)n_cam * 9 + 1), pts((size_t)n_pts * 3 + 1), uv((size_t)n_obs * 2 + 1);
        std::vector<uint64_t> row_ptr((size_t)n_cam + 1), pt_idx((size_t)n_obs + 1);
        if (n_cam) HIP_TRY(hipMemcpyAsync(bal9.data(), p->bal9, sizeof(double) * 9 * (size_t)n_cam, hipMemcpyDeviceToHost, p->stream));
        rc = c2b_problem_download(p, nullptr, pts.data(), uv.data());
        if (!rc) rc = c2b_problem_download_graph(p, row_ptr.data(), pt_idx.data());
        if (rc) return rc;
        return c2b_bal_write_as(path, 0, n_cam, bal9.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data());
    }
    rc = ensure_rows(p);
    if (rc) return rc;
    const size_t words = 3 + (size_t)n_cam + 3 * (size_t)n_obs + 9 * (size_t)n_cam + 3 * (size_t)n_pts, bytes = words * 8;
    DevBuf img;
    hipError_t e = img.alloc(bytes);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_write: %s", hipGetErrorString(e));
    uint64_t *w = img.as<uint64_t>();
    hipStream_t st = p->stream;
    // (no observations: no row structure exists and every count is zero)
    hipLaunchKernelGGL(k_bbal_rows, dim3(blocks_of(n_cam, 256)), dim3(256), 0, st, (const uint64_t *)(n_obs ? p->rows_ptr : nullptr),
                       n_cam, n_pts, n_obs, w);
    if (n_obs) hipLaunchKernelGGL(k_bbal_observations, dim3(blocks_of(n_obs, 256)), dim3(256), 0, st, (const uint32_t *)p->cam_idx,
                                  (const uint32_t *)p->pt_idx, reinterpret_cast<const double2 *>(p->uv), n_obs, w);
    uint64_t *wc = w + 3 + n_cam + 3 * (size_t)n_obs, *wp = wc + 9 * (size_t)n_cam;
    if (n_cam) hipLaunchKernelGGL(k_bbal_rows_f64, dim3(blocks_of(9 * n_cam, 256)), dim3(256), 0, st, (const double *)p->bal9, n_cam, 9, 9, wc);
    if (n_pts) hipLaunchKernelGGL(k_bbal_rows_f64, dim3(blocks_of(3 * n_pts, 256)), dim3(256), 0, st, (const double *)p->pts4, n_pts, 3, 4, wp);
    e = launch_error();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_write: %s", hipGetErrorString(e));

    const int fd = ::open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "cannot create %s", path);
    if (::ftruncate(fd, (off_t)bytes) != 0) { ::close(fd); return fail(C2B_ERR_INVALID_ARGUMENT, "write failed: %s", path); }
    // The image leaves through a ring of pinned slots: this thread copies chunk k into slot k % kSlots (26 GB/s over the
    // link), ONE writer thread pwrite()s the slots in order (8-9 GB/s into the page cache: the longer pole), so the two
    // overlap.  More writers do not help -- buffered writes to one file serialise on its inode lock -- and more threads
    // calling into the runtime cost more than they hide: measured at --blocks 128 (564 MB): 8 threads each with its own
    // pinned buffer and stream 151 ms (67 ms each just setting up), 1 thread 107 ms, this arrangement ~75 ms.
    constexpr size_t kChunk = (size_t)8 << 20;
    constexpr int kSlots = 4;
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    char *pin = nullptr;
    if (hipHostMalloc((void **)&pin, kChunk * kSlots, hipHostMallocDefault) != hipSuccess) {
        ::close(fd);
        return fail(C2B_ERR_OOM, "problem_write: no pinned staging memory");
    }
    std::mutex mu;
    std::condition_variable cv;
    size_t copied = 0, written = 0;                           // chunks copied into / written out of the ring
    int failed = 0;
    std::thread writer([&]() {
        for (size_t k = 0; k < n_chunks; ++k) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return copied > k || failed; });
                if (failed) return;
            }
            const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
            const char *src = pin + (k % kSlots) * kChunk;
            size_t done = 0;
            while (done < len) {
                const ssize_t r = ::pwrite(fd, src + done, len - done, (off_t)(off + done));
                if (r <= 0) break;
                done += (size_t)r;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (done < len) failed = 2;
            written = k + 1;
            cv.notify_all();
            if (failed) return;
        }
    });
    for (size_t k = 0; k < n_chunks; ++k) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return k < written + kSlots || failed; });      // the slot's previous chunk is on its way out
            if (failed) break;
        }
        const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
        const hipError_t ce = hipMemcpy(pin + (k % kSlots) * kChunk, reinterpret_cast<const char *>(img.ptr) + off, len, hipMemcpyDeviceToHost);
        std::lock_guard<std::mutex> lk(mu);
        if (ce != hipSuccess) failed = 1;
        copied = k + 1;
        cv.notify_all();
        if (failed) break;
    }
    writer.join();
    (void)hipHostFree(pin);
    const bool closed = ::close(fd) == 0;
    if (failed == 1) return fail(C2B_ERR_HIP, "problem_write: device-to-host copy failed");
    if (failed || !closed) return fail(C2B_ERR_INVALID_ARGUMENT, "write failed: %s", path);
    return C2B_OK;
    C2B_API_END("problem_write")
}

// BAProblem::from_file (src/baproblem.rs:697-706) into the resident problem.  `.bbal`: a reader thread streams the file
// through a ring of pinned slots, this thread sends every slot to the device as it arrives and walks the per-camera
// counts (the only part of the format that must be read in order); the per-observation decoding -- byte order, index
// range checks, the split into index and uv arrays -- and from_vec of every camera run on the device.  `.bal`: the host
// parser (decimal text is host work), then an ordinary upload.  format: -1 by extension, 0 text, 1 binary.
int c2b_problem_read(c2b_problem *p, const char *path, int format) {
    C2B_API_BEGIN
    if (!p || !path) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_read: NULL argument");
    bool binary = false;
    int rc = bal_format(path, format, &binary);
    if (rc) return rc;
    if (!binary) {
        c2b_balfile *f = nullptr;
        rc = c2b_bal_read_as(path, 0, &f);
        if (rc) return rc;
        std::unique_ptr<c2b_balfile> own(f);
        const c2b_host::Graph &g = f->g;
        return upload_common(p, g.n_cam, g.cams.data(), true, g.n_pts, g.pts.data(), g.row_ptr.data(), g.pt_idx.data(), g.uv.data());
    }
    HIP_TRY(hipSetDevice(p->device));
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "cannot open %s", path);
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    const off_t end = ::lseek(fd, 0, SEEK_END);
    if (end < 24) return fail(C2B_ERR_INVALID_ARGUMENT, "Binary parse error");
    const size_t bytes = (size_t)end & ~(size_t)7;                       // whole words (the format has nothing else)
    DevBuf raw;
    hipError_t e = raw.alloc(bytes);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
    // A few reader threads fill a ring of pinned slots (reads of one file from the page cache run in parallel, unlike
    // buffered writes), chunk k into slot k % kSlots.  The per-camera counts can only be found in order -- each sits in
    // front of its records -- so the walk over them is a chain through the whole file: the reader that has just read
    // chunk k walks the counts lying in it, as soon as chunk k - 1 has been walked, while the bytes are still in its
    // cache (walked from another thread after the fact, the 660 k dependent loads of a --blocks 128 file cost 120 ms
    // of cache misses; this way ~15).  This thread only sends walked chunks to the device, in order.
    constexpr size_t kChunk = (size_t)8 << 20;
    constexpr int kSlots = 6, kReaders = 3;
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    char *pin = nullptr;
    if (hipHostMalloc((void **)&pin, kChunk * kSlots, hipHostMallocDefault) != hipSuccess) return fail(C2B_ERR_OOM, "problem_read: no pinned staging memory");
    struct PinFree { char *q; ~PinFree() { (void)hipHostFree(q); } } pin_free{pin};
    std::mutex mu;
    std::condition_variable cv;
    size_t claimed = 0, walked = 0, drained = 0;             // chunks handed to a reader / walked / sent out of the ring
    int failed = 0;                                          // 1 copy, 2 read, 3 parse, 4 done
    auto be64 = [](const char *q) { uint64_t v; std::memcpy(&v, q, 8); return __builtin_bswap64(v); };
    uint64_t n_cam = 0, n_pts = 0, cam = 0, next_hdr = 24, n_obs = 0;       // the walk's state: owned by whoever walks chunk `walked`
    std::vector<uint64_t> row_ptr;
    auto read_loop = [&]() {
        while (true) {
            size_t k;
            {
                std::unique_lock<std::mutex> lk(mu);
                k = claimed;
                if (k >= n_chunks || failed) return;
                ++claimed;
                cv.wait(lk, [&] { return k < drained + kSlots || failed; });      // its slot's previous chunk has left
                if (failed) return;
            }
            const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
            char *dst = pin + (k % kSlots) * kChunk;
            size_t done = 0;
            while (done < len) {
                const ssize_t r = ::pread(fd, dst + done, len - done, (off_t)(off + done));
                if (r <= 0) break;
                done += (size_t)r;
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                if (done < len) failed = 2;
                cv.wait(lk, [&] { return walked == k || failed; });               // the chain reaches this chunk
                if (failed) { cv.notify_all(); return; }
            }
            bool bad = false;
            if (k == 0) {
                n_cam = be64(dst); n_pts = be64(dst + 8);    // the third word (the observation count) is not used by the reference either
                // untrusted header: a camera costs 8 + 72 bytes, a point 24 -- reject counts the file cannot hold
                if (n_cam > (bytes - 24) / 80 || n_pts > (bytes - 24) / 24 || n_cam >= ((uint64_t)1 << 32) || n_pts >= ((uint64_t)1 << 32)) bad = true;
                else row_ptr.assign((size_t)n_cam + 1, 0);
            }
            while (!bad && cam < n_cam && next_hdr < off + len) {                  // the counts whose word lies in this chunk
                const uint64_t cnt = be64(dst + (next_hdr - off));
                if (cnt > (bytes - next_hdr) / 24) { bad = true; break; }
                n_obs += cnt;
                row_ptr[(size_t)++cam] = n_obs;
                next_hdr += 8 + 24 * cnt;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (bad) failed = 3;
            walked = k + 1;
            cv.notify_all();
            if (failed) return;
        }
    };
    std::vector<std::thread> readers;
    for (int t = 0; t < (int)std::min<size_t>(kReaders, n_chunks); ++t) readers.emplace_back(read_loop);
    for (size_t k = 0; k < n_chunks; ++k) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return walked > k || failed; });
            if (failed) break;
        }
        const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
        const hipError_t ce = hipMemcpy(raw.as<char>() + off, pin + (k % kSlots) * kChunk, len, hipMemcpyHostToDevice);
        std::lock_guard<std::mutex> lk(mu);
        if (ce != hipSuccess) failed = 1;
        drained = k + 1;
        cv.notify_all();
        if (failed) break;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) failed = 4;                             // done: readers still waiting for a slot leave
    }
    cv.notify_all();
    for (auto &t : readers) t.join();
    if (failed == 1) return fail(C2B_ERR_HIP, "problem_read: host-to-device copy failed");
    if (failed == 2) return fail(C2B_ERR_INVALID_ARGUMENT, "cannot read %s", path);
    if (failed == 3 || cam < n_cam || next_hdr + 72 * n_cam + 24 * n_pts > bytes || n_obs >= ((uint64_t)1 << 32))
        return fail(C2B_ERR_INVALID_ARGUMENT, "Binary parse error");

    rc = alloc_problem(p, (int64_t)n_cam, (int64_t)n_pts, (int64_t)n_obs);
    if (rc) return rc;
    hipStream_t st = p->stream;
    DevBuf d_row, d_bad;
    e = d_row.alloc(sizeof(uint64_t) * (size_t)(n_cam + 1));
    if (e == hipSuccess) e = d_bad.alloc(4);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad.ptr, 0, 4, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_row.ptr, row_ptr.data(), sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
    rc = c2b_expand_rows(d_row.as<uint64_t>(), (int64_t)n_cam, 0, (int64_t)n_obs, p->cam_idx, st);
    if (rc) return rc;
    const uint64_t *w = raw.as<uint64_t>();
    if (n_obs) hipLaunchKernelGGL(k_bbal_read_observations, dim3(blocks_of((int64_t)n_obs, 256)), dim3(256), 0, st, w, (const uint32_t *)p->cam_idx,
                                  (int64_t)n_obs, n_pts, p->pt_idx, reinterpret_cast<double2 *>(p->uv), d_bad.as<uint32_t>());
    const uint64_t *wc = w + next_hdr / 8, *wp = wc + 9 * n_cam;
    if (n_cam) hipLaunchKernelGGL(k_bbal_read_rows_f64, dim3(blocks_of(9 * (int64_t)n_cam, 256)), dim3(256), 0, st, wc, (int64_t)n_cam, 9, 9, p->bal9);
    if (n_pts) hipLaunchKernelGGL(k_bbal_read_rows_f64, dim3(blocks_of(4 * (int64_t)n_pts, 256)), dim3(256), 0, st, wp, (int64_t)n_pts, 3, 4, p->pts4);
    e = launch_error();
    if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
    rc = c2b_cameras_from_bal(p->bal9, (int64_t)n_cam, p->cam15, st);      // SnavelyCamera::from_vec, src/baproblem.rs:180-186
    if (rc) return rc;
    uint32_t bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, d_bad.ptr, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bad) {
        free_buffers(p);
        return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "Binary parse error: point index out of range");
    }
    p->bal_valid = true;
    p->blk_valid = false;
    return C2B_OK;
    C2B_API_END("problem_read")
}

int c2b_problem_visibility_dense(c2b_problem *p, double max_dist, uint64_t *row_ptr) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_dense");
    if (!row_ptr) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense: row_ptr is NULL");
    if (!(max_dist >= 0.0)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense: max_dist must be >= 0");
    int rc = ensure_camblk(p);
    if (rc) return rc;
    free_dense(p);
    const int64_t n_tiles = c2b_visibility_dense_tiles(p->n_pts);
    const int64_t cells = p->n_cam * n_tiles;
    if (cells > ((int64_t)1 << 33))
        return fail(C2B_ERR_INVALID_ARGUMENT,
                    "problem_visibility_dense: %lld cameras x %lld point tiles is too large for the dense sweep; "
                    "use candidate pairs + c2b_problem_visibility_pairs", (long long)p->n_cam, (long long)n_tiles);
    uint32_t *d_counts = nullptr;
    uint64_t *d_tot = nullptr, *d_row = nullptr;
    hipError_t e = hipMalloc((void **)&d_counts, sizeof(uint32_t) * (size_t)(cells ? cells : 4));
    if (e == hipSuccess) e = hipMalloc((void **)&d_tot, sizeof(uint64_t) * (size_t)(p->n_cam + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_row, sizeof(uint64_t) * (size_t)(p->n_cam + 1));
    if (e == hipSuccess) {
        rc = c2b_visibility_dense_count(p->camblk, p->n_cam, p->pts4, p->n_pts, max_dist, d_counts, d_tot, d_row, p->stream);
        if (!rc) e = hipMemcpyAsync(row_ptr, d_row, sizeof(uint64_t) * (size_t)(p->n_cam + 1), hipMemcpyDeviceToHost, p->stream);
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(p->stream);
        if (!rc && e == hipSuccess) {
            const int64_t total = (int64_t)row_ptr[p->n_cam];
            e = hipMalloc((void **)&p->dense_pt, sizeof(uint32_t) * (size_t)(total ? total : 4));
            if (e == hipSuccess) e = hipMalloc((void **)&p->dense_uv, sizeof(double) * 2 * (size_t)(total ? total : 1));
            if (e == hipSuccess && total) {
                rc = c2b_visibility_dense_fill(p->camblk, p->n_cam, p->pts4, p->n_pts, max_dist, d_counts, d_row, p->dense_pt,
                                               p->dense_uv, p->stream);
                if (!rc) e = hipStreamSynchronize(p->stream);
            }
            if (!rc && e == hipSuccess) { p->dense_n = total; p->dense_row = d_row; d_row = nullptr; }
        }
    }
    if (d_counts) (void)hipFree(d_counts);
    if (d_tot) (void)hipFree(d_tot);
    if (d_row) (void)hipFree(d_row);
    if (rc || e != hipSuccess) free_dense(p);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_dense: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_visibility_dense")
}

int c2b_problem_visibility_dense_occlude(c2b_problem *p, const float *tri9, int64_t n_tri, uint64_t *row_ptr) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_dense_occlude");
    if (!p->dense_pt || !p->dense_uv || !p->dense_row)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_occlude: no sweep result");
    if (!row_ptr || n_tri < 0 || (n_tri && !tri9)) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_occlude: bad arguments");
    const int64_t n = p->dense_n, n_cam = p->n_cam;
    if (!n || !n_tri) {
        HIP_TRY(hipMemcpyAsync(row_ptr, p->dense_row, sizeof(uint64_t) * (size_t)(n_cam + 1), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        return C2B_OK;
    }
    float *d_tri = nullptr;
    uint32_t *d_cam = nullptr, *d_pt_new = nullptr, *d_flag = nullptr;
    uint32_t stack_overflow = 0;
    uint8_t *d_keep = nullptr;
    uint64_t *d_tot = nullptr, *d_row_new = nullptr;
    double *d_uv_new = nullptr;
    int rc = C2B_OK;
    // small meshes: every ray against every triangle; larger ones through a hierarchy built here on the host
    const bool use_bvh = n_tri >= kBvhMinTriangles;
    c2b_bvh *bvh = nullptr;
    void *d_nodes = nullptr;
    int64_t n_nodes = 0;
    if (use_bvh) {
        rc = c2b_bvh_build(tri9, n_tri, &bvh);
        if (rc) return rc;
        n_nodes = (int64_t)bvh->b.nodes.size();
    }
    const size_t tri_bytes = use_bvh ? (size_t)C2B_BVH_TRI_BYTES * (size_t)n_tri : sizeof(float) * 9 * (size_t)n_tri;
    const void *tri_src = use_bvh ? (const void *)bvh->b.tris.data() : (const void *)tri9;
    hipError_t e = hipMalloc((void **)&d_tri, tri_bytes);
    if (e == hipSuccess && use_bvh) e = hipMalloc(&d_nodes, (size_t)C2B_BVH_NODE_BYTES * (size_t)n_nodes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cam, sizeof(uint32_t) * (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_keep, (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_flag, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(uint32_t), p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tri, tri_src, tri_bytes, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess && use_bvh)
        e = hipMemcpyAsync(d_nodes, bvh->b.nodes.data(), (size_t)C2B_BVH_NODE_BYTES * (size_t)n_nodes, hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        rc = c2b_expand_rows(p->dense_row, n_cam, 0, n, d_cam, p->stream);
        if (!rc)
            rc = use_bvh ? c2b_occlusion_filter_bvh(p->camblk, p->pts4, d_cam, p->dense_pt, n, d_nodes, n_nodes, d_tri, n_tri, d_keep, d_flag, p->stream)
                         : c2b_occlusion_filter(p->camblk, p->pts4, d_cam, p->dense_pt, n, d_tri, n_tri, d_keep, p->stream);
        // Stable compaction of the survivor lists on the device (per-camera order of the sweep is kept): kept count per
        // row, row scan, scatter.  Only the new row pointer travels to the host.
        if (!rc) {                                         // a traversal-stack overflow invalidates the whole mask
            e = hipMemcpyAsync(&stack_overflow, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
            if (e == hipSuccess && stack_overflow)
                rc = fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_occlude: hierarchy deeper than the traversal stack");
        }
        if (!rc && e == hipSuccess) {
            int64_t w = 0;
            e = compact_rows_on_device(p, p->dense_row, d_keep, p->dense_pt, p->dense_uv, n_cam, row_ptr, &d_row_new, &d_pt_new,
                                       &d_uv_new, &w);
            if (e == hipSuccess) {                           // the filtered lists replace the sweep's
                std::swap(p->dense_pt, d_pt_new);
                std::swap(p->dense_uv, d_uv_new);
                std::swap(p->dense_row, d_row_new);
                p->dense_n = w;
            }
        }
    }
    if (rc || e != hipSuccess) (void)hipStreamSynchronize(p->stream);   // nothing below may free what a copy still reads
    if (d_tri) (void)hipFree(d_tri);
    if (d_nodes) (void)hipFree(d_nodes);
    if (d_cam) (void)hipFree(d_cam);
    if (d_keep) (void)hipFree(d_keep);
    if (d_flag) (void)hipFree(d_flag);
    if (d_tot) (void)hipFree(d_tot);
    if (d_row_new) (void)hipFree(d_row_new);
    if (d_pt_new) (void)hipFree(d_pt_new);
    if (d_uv_new) (void)hipFree(d_uv_new);
    c2b_bvh_free(bvh);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_visibility_dense_occlude: %s", hipGetErrorString(e));
    return C2B_OK;
    C2B_API_END("problem_visibility_dense_occlude")
}

int c2b_problem_visibility_dense_fetch(c2b_problem *p, uint64_t *pt_idx, double *uv) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_visibility_dense_fetch");
    if (!p->dense_pt || !p->dense_uv) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_visibility_dense_fetch: no sweep result");
    const int64_t n = p->dense_n;
    if (!n) return C2B_OK;
    if (uv) HIP_TRY(hipMemcpyAsync(uv, p->dense_uv, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, p->stream));
    if (pt_idx) {
        std::vector<uint32_t> tmp((size_t)n);
        HIP_TRY(hipMemcpyAsync(tmp.data(), p->dense_pt, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (int64_t i = 0; i < n; ++i) pt_idx[i] = tmp[(size_t)i];
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_visibility_dense_fetch")
}

static void cameras_mutated(c2b_problem *p) { p->bal_valid = false; p->blk_valid = false; }

int c2b_problem_add_drift(c2b_problem *p, double strength, double angle_strength, double std, const double dir[3],
                          uint64_t seed) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_drift");
    if (!dir) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_drift: dir is NULL");
    int rc = compute_stats(p);
    if (rc) return rc;
    rc = c2b_add_drift(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats + 15, strength, angle_strength, std, dir[0],
                       dir[1], dir[2], seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_drift")
}

int c2b_problem_add_drift_normalized(c2b_problem *p, double strength, double angle_strength, double std,
                                     uint64_t seed) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_drift_normalized");
    int rc = compute_stats(p);
    if (rc) return rc;
    rc = c2b_add_drift_normalized(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, strength, angle_strength, std, seed,
                                  p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_drift_normalized")
}

int c2b_problem_add_noise(c2b_problem *p, double translation_std, double rotation_std, double point_std,
                          double observations_std, uint64_t seed) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_noise");
    int rc = compute_stats(p);
    if (rc) return rc;
    rc = c2b_add_noise_entities(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, translation_std, rotation_std,
                                point_std, seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    rc = c2b_add_noise_observations(p->uv, p->n_obs, 0, observations_std, seed, p->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_noise")
}

// add_noise followed by the L1 / L2 errors of the result -- run_noise's tail (src/bin/city2ba.rs:334-354) -- with the
// observation pass and both error sums in one launch.  comm != NULL: the problem is a shard (statistics and the
// 2-element sum go through the communicator).
static int sharded_stats(c2b_problem *p, c2b_comm *comm);
static int add_noise_errors_impl(c2b_problem *p, c2b_comm *comm, double translation_std, double rotation_std, double point_std,
                                 double observations_std, uint64_t seed, double *l1, double *l2) {
    int rc = comm ? sharded_stats(p, comm) : compute_stats(p);
    if (rc) return rc;
    rc = comm ? c2b_add_noise_entities_sharded(p->cam15, p->n_cam, p->shard_cam_base, p->pts4, p->n_pts, p->stats, translation_std,
                                               rotation_std, point_std, seed, p->stream)
              : c2b_add_noise_entities(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, translation_std, rotation_std, point_std,
                                       seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    rc = ensure_camblk(p);                                 // the perturbed cameras' records
    if (!rc) rc = ensure_rows(p);
    if (rc) return rc;
    if (p->n_obs > 0)
        rc = c2b_add_noise_observations_error_sums2_rows(p->camblk, p->pts4, p->rows_ptr, p->n_cam, p->rows_tiles, p->pt_idx, p->uv,
                                                         p->n_obs, comm ? p->shard_obs_base : 0, observations_std, seed, p->ws,
                                                         p->scalar, p->stream);
    else
        HIP_TRY(hipMemsetAsync(p->scalar, 0, 2 * sizeof(double), p->stream));
    if (!rc && comm) rc = c2b_comm_all_reduce_sum_f64(comm, p->scalar, 2, p->stream);
    if (rc) return rc;
    double sums[2] = {0.0, 0.0};
    HIP_TRY(hipMemcpyAsync(sums, p->scalar, 2 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *l1 = std::pow(sums[0], 1.0 / 1.0);
    *l2 = std::pow(sums[1], 1.0 / 2.0);
    return C2B_OK;
}

int c2b_problem_add_noise_errors_l1_l2(c2b_problem *p, double translation_std, double rotation_std, double point_std,
                                       double observations_std, uint64_t seed, double *l1, double *l2) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_noise_errors_l1_l2");
    if (!l1 || !l2) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_noise_errors_l1_l2: NULL output");
    return add_noise_errors_impl(p, nullptr, translation_std, rotation_std, point_std, observations_std, seed, l1, l2);
    C2B_API_END("problem_add_noise_errors_l1_l2")
}

int c2b_problem_add_sin_noise(c2b_problem *p, const double dir[3], const double noise_dir[3], double strength,
                              double frequency) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_add_sin_noise");
    if (!dir || !noise_dir) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_sin_noise: NULL direction");
    int rc = compute_stats(p);
    if (rc) return rc;
    rc = c2b_add_sin_noise(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, dir[0], dir[1], dir[2], noise_dir[0],
                           noise_dir[1], noise_dir[2], strength, frequency, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_sin_noise")
}

// ---- Level 1 for a problem that is ONE SHARD of a larger one -------------------------------
// One c2b_problem per GPU holds a contiguous camera range (c2b_partition_cameras), its slice of the observation list
// and the WHOLE point table.  After c2b_problem_set_shard the *_sharded entries below give, shard by shard, exactly what
// the unsharded calls give on the whole problem: draws are keyed by global indices, the statistics go through the
// communicator (c2b_stats_sharded), every rank perturbs the replicated points identically.  All are collective (every
// rank of the communicator calls them in the same order) and synchronous.
int c2b_problem_set_shard(c2b_problem *p, int64_t cam_base, int64_t n_cam_global, int64_t obs_base) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_set_shard");
    if (cam_base < 0 || obs_base < 0 || n_cam_global < cam_base + p->n_cam)
        return fail(C2B_ERR_INVALID_ARGUMENT, "problem_set_shard: the shard [%lld, %lld) does not fit %lld cameras",
                    (long long)cam_base, (long long)(cam_base + p->n_cam), (long long)n_cam_global);
    p->shard_cam_base = cam_base; p->shard_n_cam_global = n_cam_global; p->shard_obs_base = obs_base;
    return C2B_OK;
    C2B_API_END("problem_set_shard")
}

#define NEED_SHARD(p, comm, who)                                                                          \
    NEED_UPLOADED(p, who);                                                                                \
    if (!(comm)) return fail(C2B_ERR_INVALID_ARGUMENT, who ": communicator is NULL");                     \
    if ((p)->shard_n_cam_global < 0) return fail(C2B_ERR_INVALID_ARGUMENT, who ": c2b_problem_set_shard first"); \
    if ((comm)->device != (p)->device) return fail(C2B_ERR_INVALID_ARGUMENT, who ": communicator and problem live on different devices")

static int sharded_stats(c2b_problem *p, c2b_comm *comm) {
    int rc = ensure_camblk(p);
    if (rc) return rc;
    return c2b_stats_sharded(comm, p->camblk, p->n_cam, p->shard_cam_base, p->shard_n_cam_global, p->pts4, p->n_pts, p->ws,
                             p->stats, p->stream);
}

int c2b_problem_stats_sharded(c2b_problem *p, c2b_comm *comm, double *stats) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_stats_sharded");
    if (!stats) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_stats_sharded: stats is NULL");
    const int rc = sharded_stats(p, comm);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(stats, p->stats, sizeof(double) * C2B_STATS_DOUBLES, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_stats_sharded")
}

// dir == NULL: add_drift_normalized (direction and scale from the global std, src/noise.rs:47-56)
int c2b_problem_add_drift_sharded(c2b_problem *p, c2b_comm *comm, double strength, double angle_strength, double std,
                                  const double *dir, uint64_t seed) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_add_drift_sharded");
    int rc = sharded_stats(p, comm);
    if (rc) return rc;
    rc = c2b_add_drift_sharded(p->cam15, p->n_cam, p->shard_cam_base, p->pts4, p->n_pts, p->stats, dir ? 0 : 1, strength,
                               angle_strength, std, dir ? dir[0] : 0.0, dir ? dir[1] : 0.0, dir ? dir[2] : 0.0, seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_drift_sharded")
}

int c2b_problem_add_noise_sharded(c2b_problem *p, c2b_comm *comm, double translation_std, double rotation_std,
                                  double point_std, double observations_std, uint64_t seed) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_add_noise_sharded");
    int rc = sharded_stats(p, comm);
    if (rc) return rc;
    rc = c2b_add_noise_entities_sharded(p->cam15, p->n_cam, p->shard_cam_base, p->pts4, p->n_pts, p->stats, translation_std,
                                        rotation_std, point_std, seed, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    rc = c2b_add_noise_observations(p->uv, p->n_obs, p->shard_obs_base, observations_std, seed, p->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_noise_sharded")
}

int c2b_problem_add_noise_errors_l1_l2_sharded(c2b_problem *p, c2b_comm *comm, double translation_std, double rotation_std,
                                               double point_std, double observations_std, uint64_t seed, double *l1, double *l2) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_add_noise_errors_l1_l2_sharded");
    if (!l1 || !l2) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_noise_errors_l1_l2_sharded: NULL output");
    return add_noise_errors_impl(p, comm, translation_std, rotation_std, point_std, observations_std, seed, l1, l2);
    C2B_API_END("problem_add_noise_errors_l1_l2_sharded")
}

int c2b_problem_add_sin_noise_sharded(c2b_problem *p, c2b_comm *comm, const double dir[3], const double noise_dir[3],
                                      double strength, double frequency) {
    C2B_API_BEGIN
    NEED_SHARD(p, comm, "problem_add_sin_noise_sharded");
    if (!dir || !noise_dir) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_add_sin_noise_sharded: NULL direction");
    int rc = sharded_stats(p, comm);                        // the extent of the WHOLE problem scales the phase
    if (rc) return rc;
    rc = c2b_add_sin_noise(p->cam15, p->n_cam, p->pts4, p->n_pts, p->stats, dir[0], dir[1], dir[2], noise_dir[0],
                           noise_dir[1], noise_dir[2], strength, frequency, p->stream);
    if (rc) return rc;
    cameras_mutated(p);
    HIP_TRY(hipStreamSynchronize(p->stream));
    return C2B_OK;
    C2B_API_END("problem_add_sin_noise_sharded")
}

}  // extern "C"
