// capi_files.hpp -- BAProblem::write / from_file of the RESIDENT problem (c2b_problem_write, c2b_problem_read): the .bbal
// words (cell_kernels.hpp: k_bbal_*) and the .bal decimal text (text_kernels.hpp + decimal.hpp) are assembled / taken
// apart on the device; the host only moves bytes between the file and pinned memory (image_to_file, file_to_device).
// Part of the one translation unit of the C ABI: included by capi_problem.hpp, never compiled or included on its own.
// decimal.hpp's tables on the device: computed once on the host, copied once per device
static int device_dec_tables(int device, const c2b_dec::Tables **out) {
    static std::mutex mu;
    static const c2b_dec::Tables *on_device[64] = {};
    if (device < 0 || device >= 64) return fail(C2B_ERR_INVALID_ARGUMENT, "device %d out of range", device);
    std::lock_guard<std::mutex> lk(mu);
    if (!on_device[device]) {
        HIP_TRY(hipSetDevice(device));
        void *addr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&addr, HIP_SYMBOL(g_dec_tables)));
        HIP_TRY(hipMemcpy(addr, &c2b_dec::host_tables(), sizeof(c2b_dec::Tables), hipMemcpyHostToDevice));
        on_device[device] = static_cast<const c2b_dec::Tables *>(addr);
    }
    *out = on_device[device];
    return C2B_OK;
}

// A device-resident file image -> `path`.  The image leaves through a ring of pinned slots: this thread copies chunk k
// into slot k % kSlots (26 GB/s over the link), ONE writer thread pwrite()s the slots in order (8-9 GB/s into the page
// cache: the longer pole), so the two overlap.  More writers do not help -- buffered writes to one file serialise on its
// inode lock -- and more threads calling into the runtime cost more than they hide: measured at --blocks 128 (564 MB):
// 8 threads each with its own pinned buffer and stream 151 ms (67 ms each just setting up), 1 thread 107 ms, this
// arrangement ~75 ms.
static int image_to_file(const char *path, const void *dev, size_t bytes) {
    const int fd = ::open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) return fail(C2B_ERR_INVALID_ARGUMENT, "cannot create %s", path);
    if (::ftruncate(fd, (off_t)bytes) != 0) { ::close(fd); return fail(C2B_ERR_INVALID_ARGUMENT, "write failed: %s", path); }
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
    auto write_loop = [&]() {
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
    };
    std::thread writer;
    try {
        writer = std::thread(write_loop);
    } catch (...) {
        (void)hipHostFree(pin);
        ::close(fd);
        return fail(C2B_ERR_OOM, "problem_write: cannot start the writer thread");
    }
    for (size_t k = 0; k < n_chunks; ++k) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return k < written + kSlots || failed; });      // the slot's previous chunk is on its way out
            if (failed) break;
        }
        const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
        const hipError_t ce = hipMemcpy(pin + (k % kSlots) * kChunk, static_cast<const char *>(dev) + off, len, hipMemcpyDeviceToHost);
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
}

// BAProblem::write (src/baproblem.rs:768-785) of the RESIDENT problem.  `.bbal` (format 1): the file image is assembled
// on the device (cell_kernels.hpp: k_bbal_*: to_vec of every camera, the per-camera counts, the byte order) and leaves
// through a ring of pinned slots (image_to_file) -- the host touches no observation.  `.bal` (format 0): the same, the
// image being text (text_kernels.hpp: shortest round-trip decimals on the device; options.host_text = the host formatter of
// csrc/host_baproblem.hpp over a download, the same bytes).  format -1: by extension, like the reference.
int c2b_problem_write(c2b_problem *p, const char *path, int format) {
    C2B_API_BEGIN
    NEED_UPLOADED(p, "problem_write");
    if (!path) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_write: path is NULL");
    bool binary = false;
    int rc = bal_format(path, format, &binary);
    if (rc) return rc;
    const int64_t n_cam = p->n_cam, n_pts = p->n_pts, n_obs = p->n_obs;
    if (!p->bal_valid && !p->bal9_fresh) {                   // to_vec (src/baproblem.rs:189-202) of the current state
        rc = c2b_cameras_to_bal(p->cam15, n_cam, p->bal9, p->stream);
        if (rc) return rc;
        // bal9 now holds to_vec of the cameras: the next write / download of the unmodified problem skips the pass.  NOT
        // bal_valid: that flag makes bal9 the truth (R re-derived as from_rodrigues(w), last bits apart from the state's
        // R), and writing a file must not move a single projection.
        p->bal9_fresh = true;
    }
    const c2b_host::IoThreadsScope io_scope(p->opt.io_threads);
    if (!binary && p->opt.host_text) {                       // the host formatter over a download (rounds 1-3's route)
        std::vector<double> bal9((size_t)n_cam * 9 + 1), pts((size_t)n_pts * 3 + 1), uv((size_t)n_obs * 2 + 1);
        std::vector<uint64_t> row_ptr((size_t)n_cam + 1), pt_idx((size_t)n_obs + 1);
        if (n_cam) HIP_TRY(hipMemcpyAsync(bal9.data(), p->bal9, sizeof(double) * 9 * (size_t)n_cam, hipMemcpyDeviceToHost, p->stream));
        rc = c2b_problem_download(p, nullptr, pts.data(), uv.data());
        if (!rc) rc = c2b_problem_download_graph(p, row_ptr.data(), pt_idx.data());
        if (rc) return rc;
        return c2b_bal_write_as(path, 0, n_cam, bal9.data(), n_pts, pts.data(), row_ptr.data(), pt_idx.data(), uv.data());
    }
    if (!binary) {
        // text_kernels.hpp: tile byte counts, 64-bit tile bases, then the characters.  Units: one observation line, or one
        // value of a camera / point line with the separator behind it.
        const c2b_dec::Tables *T = nullptr;
        rc = device_dec_tables(p->device, &T);
        if (rc) return rc;
        hipStream_t st = p->stream;
        const int64_t t_obs = (n_obs + kTextTile - 1) / kTextTile, t_cam = (9 * n_cam + kTextTile - 1) / kTextTile,
                      t_pts = (3 * n_pts + kTextTile - 1) / kTextTile, n_tiles = t_obs + t_cam + t_pts;
        char head[80];
        const int head_len = std::snprintf(head, sizeof head, "%lld %lld %lld\n", (long long)n_cam, (long long)n_pts, (long long)n_obs);
        DevBuf tile_len, tile_base, total;
        hipError_t e = tile_len.alloc(4 * (size_t)(n_tiles + 1));
        if (e == hipSuccess) e = tile_base.alloc(8 * (size_t)(n_tiles + 1));
        if (e == hipSuccess) e = total.alloc(8);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_write: %s", hipGetErrorString(e));
        uint32_t *tl = tile_len.as<uint32_t>();
        uint64_t *tb = tile_base.as<uint64_t>();
        const uint32_t *ci = (const uint32_t *)p->cam_idx, *pi = (const uint32_t *)p->pt_idx;
        const double2 *uvd = reinterpret_cast<const double2 *>(p->uv);
        if (t_obs) hipLaunchKernelGGL((k_text_obs<false>), dim3((unsigned)t_obs), dim3(kTextTile), 0, st, ci, pi, uvd, n_obs, T, tl,
                                      (const uint64_t *)nullptr, (char *)nullptr);
        if (t_cam) hipLaunchKernelGGL((k_text_vals<false>), dim3((unsigned)t_cam), dim3(kTextTile), 0, st, (const double *)p->bal9, n_cam, 9, 9, T,
                                      tl + t_obs, (const uint64_t *)nullptr, (char *)nullptr);
        if (t_pts) hipLaunchKernelGGL((k_text_vals<false>), dim3((unsigned)t_pts), dim3(kTextTile), 0, st, (const double *)p->pts4, n_pts, 3, 4, T,
                                      tl + t_obs + t_cam, (const uint64_t *)nullptr, (char *)nullptr);
        hipLaunchKernelGGL(k_text_tile_bases, dim3(1), dim3(1024), 0, st, (const uint32_t *)tl, n_tiles, (uint64_t)head_len, tb, total.as<uint64_t>());
        e = launch_error();
        uint64_t bytes = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&bytes, total.ptr, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        DevBuf img;
        if (e == hipSuccess) e = img.alloc((size_t)bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(img.ptr, head, (size_t)head_len, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_write: %s", hipGetErrorString(e));
        char *im = img.as<char>();
        if (t_obs) hipLaunchKernelGGL((k_text_obs<true>), dim3((unsigned)t_obs), dim3(kTextTile), 0, st, ci, pi, uvd, n_obs, T, (uint32_t *)nullptr,
                                      (const uint64_t *)tb, im);
        if (t_cam) hipLaunchKernelGGL((k_text_vals<true>), dim3((unsigned)t_cam), dim3(kTextTile), 0, st, (const double *)p->bal9, n_cam, 9, 9, T,
                                      (uint32_t *)nullptr, (const uint64_t *)(tb + t_obs), im);
        if (t_pts) hipLaunchKernelGGL((k_text_vals<true>), dim3((unsigned)t_pts), dim3(kTextTile), 0, st, (const double *)p->pts4, n_pts, 3, 4, T,
                                      (uint32_t *)nullptr, (const uint64_t *)(tb + t_obs + t_cam), im);
        e = launch_error();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_write: %s", hipGetErrorString(e));
        return image_to_file(path, img.ptr, (size_t)bytes);
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

    return image_to_file(path, img.ptr, bytes);
    C2B_API_END("problem_write")
}

// decimal.hpp's reading tables on the device, like device_dec_tables
static int device_parse_tables(int device, const c2b_dec::ParseTables **out) {
    static std::mutex mu;
    static const c2b_dec::ParseTables *on_device[64] = {};
    if (device < 0 || device >= 64) return fail(C2B_ERR_INVALID_ARGUMENT, "device %d out of range", device);
    std::lock_guard<std::mutex> lk(mu);
    if (!on_device[device]) {
        HIP_TRY(hipSetDevice(device));
        void *addr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&addr, HIP_SYMBOL(g_parse_tables)));
        HIP_TRY(hipMemcpy(addr, &c2b_dec::host_parse_tables(), sizeof(c2b_dec::ParseTables), hipMemcpyHostToDevice));
        on_device[device] = static_cast<const c2b_dec::ParseTables *>(addr);
    }
    *out = on_device[device];
    return C2B_OK;
}

// `bytes` of an open file -> device memory through a ring of pinned slots: a few reader threads pread() chunk k into
// slot k % kSlots (reads of one file from the page cache run in parallel, unlike buffered writes), this thread sends the
// slots to the device in order.  0 = ok, 1 = copy failed, 2 = read failed, 3 = no resources.
static int file_to_device(int fd, size_t bytes, char *dev, int read_threads = 0) {
    constexpr size_t kChunk = (size_t)8 << 20;
    int kSlots = 6, kReaders = 3;
    if (read_threads > 0) { kReaders = read_threads; kSlots = 2 * kReaders; }
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    char *pin = nullptr;
    if (hipHostMalloc((void **)&pin, kChunk * kSlots, hipHostMallocDefault) != hipSuccess) return 3;
    struct PinFree { char *q; ~PinFree() { (void)hipHostFree(q); } } pin_free{pin};
    std::mutex mu;
    std::condition_variable cv;
    size_t claimed = 0, drained = 0;
    std::vector<char> ready(n_chunks, 0);
    int failed = 0;
    auto read_loop = [&]() {
        while (true) {
            size_t k;
            {
                std::unique_lock<std::mutex> lk(mu);
                k = claimed;
                if (k >= n_chunks || failed) return;
                ++claimed;
                cv.wait(lk, [&] { return k < drained + (size_t)kSlots || failed; });      // its slot's previous chunk has left
                if (failed) return;
            }
            const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
            char *dst = pin + (k % (size_t)kSlots) * kChunk;
            size_t done = 0;
            while (done < len) {
                const ssize_t r = ::pread(fd, dst + done, len - done, (off_t)(off + done));
                if (r <= 0) break;
                done += (size_t)r;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (done < len) failed = 2;
            ready[k] = 1;
            cv.notify_all();
            if (failed) return;
        }
    };
    std::vector<std::thread> readers;
    readers.reserve(kReaders);
    try {
        for (int t = 0; t < (int)std::min<size_t>(kReaders, n_chunks); ++t) readers.emplace_back(read_loop);
    } catch (...) {
        { std::lock_guard<std::mutex> lk(mu); failed = 3; }
        cv.notify_all();
        for (auto &t : readers) t.join();
        return 3;
    }
    for (size_t k = 0; k < n_chunks; ++k) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready[k] || failed; });
            if (failed) break;
        }
        const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
        const hipError_t ce = hipMemcpy(dev + off, pin + (k % kSlots) * kChunk, len, hipMemcpyHostToDevice);
        std::lock_guard<std::mutex> lk(mu);
        if (ce != hipSuccess) failed = 1;
        drained = k + 1;
        cv.notify_all();
        if (failed) break;
    }
    int outcome;
    {
        std::lock_guard<std::mutex> lk(mu);
        outcome = failed;
        if (!failed) failed = 4;                             // done: readers still waiting for a slot leave
    }
    cv.notify_all();
    for (auto &t : readers) t.join();
    return outcome;
}

// from_file_text (src/baproblem.rs:580-629) on the device (text_kernels.hpp).  *handled = false: the device declined --
// a small file, a spelling or a digit count decimal.hpp leaves to strtod, counts that do not fit the file, an index out
// of range -- and the caller runs the host parser, which owns every corner of the
// grammar and the wording of every error.  The problem is replaced only after the whole file has parsed.
static int read_text_device(c2b_problem *p, const char *path, bool *handled) {
    *handled = false;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return C2B_OK;                                // the host path words the error
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    const off_t end = ::lseek(fd, 0, SEEK_END);
    const size_t min_bytes = p->opt.text_device_min_bytes >= 0 ? (size_t)p->opt.text_device_min_bytes : (size_t)64 << 10;
    if (end < 6 || (size_t)end < min_bytes || (uint64_t)end >= ((uint64_t)1 << 32)) return C2B_OK;
    const size_t bytes = (size_t)end;
    // the three counts, from the first bytes
    char head[256];
    const ssize_t got = ::pread(fd, head, sizeof head, 0);
    if (got <= 0) return C2B_OK;
    uint64_t hdr[3];
    {
        auto ws = [](char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; };
        ssize_t i = 0;
        for (int k = 0; k < 3; ++k) {
            while (i < got && ws(head[i])) ++i;
            const ssize_t b = i;
            uint64_t v = 0;
            while (i < got && head[i] >= '0' && head[i] <= '9' && i - b < 19) v = v * 10 + (uint64_t)(head[i++] - '0');
            if (i == b || i >= got || !ws(head[i])) return C2B_OK;
            hdr[k] = v;
        }
    }
    const uint64_t nc = hdr[0], np = hdr[1], no = hdr[2];
    if (no > bytes / 8 || nc > bytes / 18 || np > bytes / 6 || nc >= ((uint64_t)1 << 32) || np >= ((uint64_t)1 << 32) || no >= ((uint64_t)1 << 31))
        return C2B_OK;
    HIP_TRY(hipSetDevice(p->device));
    const c2b_dec::ParseTables *T = nullptr;
    int rc = device_parse_tables(p->device, &T);
    if (rc) return rc;
    hipStream_t st = p->stream;
    const size_t padded = ((bytes + 15) & ~(size_t)15) + 16;
    const int64_t n_tiles = (int64_t)((bytes + kParseTile - 1) / kParseTile);
    DevBuf raw, cnt, base, total, flags, t_cam, t_pt, t_uv, t_bal, t_pts;
    hipError_t e = raw.alloc(padded);
    if (e == hipSuccess) e = cnt.alloc(4 * (size_t)n_tiles);
    if (e == hipSuccess) e = base.alloc(8 * (size_t)n_tiles);
    if (e == hipSuccess) e = total.alloc(8);
    if (e == hipSuccess) e = flags.alloc(16);
    if (e == hipSuccess) e = t_cam.alloc(4 * (size_t)no);
    if (e == hipSuccess) e = t_pt.alloc(4 * (size_t)no);
    if (e == hipSuccess) e = t_uv.alloc(16 * (size_t)no);
    if (e == hipSuccess) e = t_bal.alloc(72 * (size_t)nc);
    if (e == hipSuccess) e = t_pts.alloc(32 * (size_t)np);
    if (e == hipSuccess) e = hipMemsetAsync(raw.as<char>() + (padded - 32), 0, 32, st);
    if (e == hipSuccess) e = hipMemsetAsync(flags.ptr, 0, 16, st);
    if (e == hipSuccess) e = hipMemsetAsync(t_pts.ptr, 0, np ? 32 * (size_t)np : 16, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
    const int io = file_to_device(fd, bytes, raw.as<char>(), p->opt.read_threads);
    if (io == 1) return fail(C2B_ERR_HIP, "problem_read: host-to-device copy failed");
    if (io) return C2B_OK;                                   // unreadable: the host path says so
    hipLaunchKernelGGL(k_text_count_tokens, dim3((unsigned)n_tiles), dim3(kTextTile), 0, st, (const char *)raw.as<char>(), (int64_t)bytes, cnt.as<uint32_t>());
    hipLaunchKernelGGL(k_text_tile_bases, dim3(1), dim3(1024), 0, st, (const uint32_t *)cnt.as<uint32_t>(), n_tiles, (uint64_t)0, base.as<uint64_t>(),
                       total.as<uint64_t>());
    e = launch_error();
    uint64_t n_tokens = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n_tokens, total.ptr, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
    if (n_tokens < 3 + 4 * no + 9 * nc + 3 * np) return C2B_OK;           // too short: the host parser says where
    hipLaunchKernelGGL(k_text_parse, dim3((unsigned)n_tiles), dim3(kTextTile), 0, st, (const char *)raw.as<char>(), (int64_t)bytes,
                       (const uint64_t *)base.as<uint64_t>(), T, nc, np, no, t_cam.as<uint32_t>(), t_pt.as<uint32_t>(), t_uv.as<double>(),
                       t_bal.as<double>(), t_pts.as<double>(), flags.as<uint32_t>());
    if (no > 1) hipLaunchKernelGGL(k_text_check_sorted, dim3(blocks_of((int64_t)no, 256)), dim3(256), 0, st, (const uint32_t *)t_cam.as<uint32_t>(), (int64_t)no,
                                   flags.as<uint32_t>());
    e = launch_error();
    uint32_t fl[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(fl, flags.ptr, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
    if (fl[0] || fl[1] || fl[2]) return C2B_OK;
    DevBuf s_cam, s_pt, s_uv;                               // the camera-major lists when the file's are not
    if (fl[3]) {
        // BAProblem::new's per-camera push in file order = a stable sort by camera (text_kernels.hpp: k_sort_*): 8 bits
        // a pass over (camera, position in the file), then one gather of the point indices and the observations
        const int64_t n = (int64_t)no, n_st = (n + kSortTile - 1) / kSortTile, n_hist = 256 * n_st;
        DevBuf k2, v1, v2, hist, offs, tiles, total32;
        e = k2.alloc(4 * (size_t)n);
        if (e == hipSuccess) e = v1.alloc(4 * (size_t)n);
        if (e == hipSuccess) e = v2.alloc(4 * (size_t)n);
        if (e == hipSuccess) e = hist.alloc(4 * (size_t)n_hist);
        if (e == hipSuccess) e = offs.alloc(4 * (size_t)n_hist);
        if (e == hipSuccess) e = tiles.alloc(4 * (size_t)(n_hist / kScanTile + 2));
        if (e == hipSuccess) e = total32.alloc(4);
        if (e == hipSuccess) e = s_cam.alloc(4 * (size_t)n);
        if (e == hipSuccess) e = s_pt.alloc(4 * (size_t)n);
        if (e == hipSuccess) e = s_uv.alloc(16 * (size_t)n);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? C2B_ERR_OOM : C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
        int bits = 1;
        while (bits < 32 && (nc - 1) >> bits) ++bits;
        // keys ping-pong between t_cam / k2 so that the last pass lands in s_cam; values between v1 / v2
        const int passes = (bits + 7) / 8;
        const uint32_t *k_in = t_cam.as<uint32_t>(), *v_in = nullptr;
        for (int ps = 0; ps < passes; ++ps) {
            uint32_t *k_out = ps == passes - 1 ? s_cam.as<uint32_t>() : (ps % 2 == 0 ? k2.as<uint32_t>() : t_cam.as<uint32_t>());
            uint32_t *v_out = ps % 2 == 0 ? v1.as<uint32_t>() : v2.as<uint32_t>();
            hipLaunchKernelGGL(k_sort_hist, dim3((unsigned)n_st), dim3(64), 0, st, k_in, n, 8 * ps, n_st, hist.as<uint32_t>());
            uint32_t sum = 0;
            e = scan_flags(st, hist.as<uint32_t>(), n_hist, offs.as<uint32_t>(), tiles.as<uint32_t>(), total32.as<uint32_t>(), &sum);
            if (e == hipSuccess && (int64_t)sum != n) return fail(C2B_ERR_HIP, "problem_read: the sort's histogram does not add up");
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)n_st), dim3(64), 0, st, k_in, v_in, n, 8 * ps, n_st, (const uint32_t *)offs.as<uint32_t>(),
                                   k_out, v_out);
                e = launch_error();
            }
            if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
            k_in = k_out; v_in = v_out;
        }
        hipLaunchKernelGGL(k_text_gather_obs, dim3(blocks_of(n, 256)), dim3(256), 0, st, v_in, n, (const uint32_t *)t_pt.as<uint32_t>(),
                           (const double2 *)t_uv.as<double2>(), s_pt.as<uint32_t>(), s_uv.as<double2>());
        e = launch_error();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e));
    }
    const void *f_cam = fl[3] ? s_cam.ptr : t_cam.ptr, *f_pt = fl[3] ? s_pt.ptr : t_pt.ptr, *f_uv = fl[3] ? s_uv.ptr : t_uv.ptr;
    rc = alloc_problem(p, (int64_t)nc, (int64_t)np, (int64_t)no);
    if (rc) return rc;
    if (no) e = hipMemcpyAsync(p->cam_idx, f_cam, 4 * (size_t)no, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && no) e = hipMemcpyAsync(p->pt_idx, f_pt, 4 * (size_t)no, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && no) e = hipMemcpyAsync(p->uv, f_uv, 16 * (size_t)no, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && nc) e = hipMemcpyAsync(p->bal9, t_bal.ptr, 72 * (size_t)nc, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && np) e = hipMemcpyAsync(p->pts4, t_pts.ptr, 32 * (size_t)np, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) { free_buffers(p); return fail(C2B_ERR_HIP, "problem_read: %s", hipGetErrorString(e)); }
    rc = c2b_cameras_from_bal(p->bal9, (int64_t)nc, p->cam15, st);         // SnavelyCamera::from_vec, src/baproblem.rs:180-186
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail(C2B_ERR_HIP, "problem_read: synchronize failed");
    if (rc) { free_buffers(p); return rc; }
    p->bal_valid = true;
    p->blk_valid = false;
    *handled = true;
    return C2B_OK;
}

// BAProblem::from_file (src/baproblem.rs:697-706) into the resident problem.  `.bbal`: a reader thread streams the file
// through a ring of pinned slots, this thread sends every slot to the device as it arrives and walks the per-camera
// counts (the only part of the format that must be read in order); the per-observation decoding -- byte order, index
// range checks, the split into index and uv arrays -- and from_vec of every camera run on the device.  `.bal`: tokenised
// and parsed on the device (read_text_device above); whatever that declines goes through the host parser and an ordinary
// upload (options.host_text: always).  format: -1 by extension, 0 text, 1 binary.
int c2b_problem_read(c2b_problem *p, const char *path, int format) {
    C2B_API_BEGIN
    if (!p || !path) return fail(C2B_ERR_INVALID_ARGUMENT, "problem_read: NULL argument");
    bool binary = false;
    int rc = bal_format(path, format, &binary);
    if (rc) return rc;
    const c2b_host::IoThreadsScope io_scope(p->opt.io_threads);
    if (!binary && !p->opt.host_text) {
        bool handled = false;
        rc = read_text_device(p, path, &handled);
        if (rc || handled) return rc;
        if (p->opt.text_device_strict)                       // tests: make sure the device path is the one that ran
            return fail(C2B_ERR_INVALID_ARGUMENT, "problem_read: the device parser declined %s", path);
    }
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
    readers.reserve(kReaders);
    try {
        for (int t = 0; t < (int)std::min<size_t>(kReaders, n_chunks); ++t) readers.emplace_back(read_loop);
    } catch (...) {                                          // no more threads to be had: the ones started must be joined
        {
            std::lock_guard<std::mutex> lk(mu);
            failed = 2;
        }
        cv.notify_all();
        for (auto &t : readers) t.join();
        return fail(C2B_ERR_OOM, "problem_read: cannot start a reader thread");
    }
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
    // From here on the handle owns freshly allocated, still undecoded arrays: every way out but the last one drops them,
    // so that a failed read leaves "nothing uploaded" behind and never a problem that passes NEED_UPLOADED with garbage.
    struct DropUnlessDone { c2b_problem *q; bool done = false; ~DropUnlessDone() { if (!done) { (void)hipStreamSynchronize(q->stream); free_buffers(q); } } } guard{p};
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
    if (bad) return fail(C2B_ERR_INDEX_OUT_OF_RANGE, "Binary parse error: point index out of range");
    guard.done = true;
    p->bal_valid = true;
    p->blk_valid = false;
    return C2B_OK;
    C2B_API_END("problem_read")
}
