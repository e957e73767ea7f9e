// Host file I/O of the container codec (the .b files of conduct_encoding / conduct_decoding) on plain C++
// threads: a list of (path, file offset, byte count, memory address) ranges is read or written by `threads` workers that
// take ranges from a shared counter.  The Python driver used eight interpreter threads calling os.preadv / os.pwrite for
// this; every piece they finish makes them queue for the GIL, and the decoder's main thread — which at that moment runs
// the Python-heavy prologue (checkpoint, prior tables, level plan) — ran 3-10 x slower beside them
// (tools/ckpt_load_micro.py).  One ctypes call (the GIL is released for its whole duration) per group of ranges instead.
#include <atomic>
#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "cgs_internal.h"

namespace {

template <bool WRITE>
int run_ranges(int n, const char *const *paths, const int64_t *file_off, const int64_t *nbytes, void *const *mem, int threads) {
    if (n < 0 || (n > 0 && (!paths || !file_off || !nbytes || !mem))) { cgs_set_error("file ranges: bad arguments"); return CGS_ERR_ARG; }
    if (n == 0) return CGS_OK;
    std::atomic<int> next(0);
    std::atomic<bool> failed(false);
    std::mutex err_lock;
    std::string err;
    auto fail = [&](const std::string &what) {
        std::lock_guard<std::mutex> g(err_lock);
        if (!failed.exchange(true)) err = what;
    };
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || failed.load()) return;
            if (nbytes[i] < 0 || file_off[i] < 0 || !paths[i] || (nbytes[i] > 0 && !mem[i])) { fail("bad range"); return; }
            const int fd = WRITE ? open(paths[i], O_WRONLY | O_CREAT, 0644) : open(paths[i], O_RDONLY);
            if (fd < 0) { fail(std::string(paths[i]) + ": " + strerror(errno)); return; }
            int64_t done = 0;
            while (done < nbytes[i]) {
                const ssize_t got = WRITE ? pwrite(fd, (const char *)mem[i] + done, (size_t)(nbytes[i] - done), (off_t)(file_off[i] + done))
                                          : pread(fd, (char *)mem[i] + done, (size_t)(nbytes[i] - done), (off_t)(file_off[i] + done));
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) {
                    fail(std::string(paths[i]) + (got == 0 ? ": file shorter than the range" : std::string(": ") + strerror(errno)));
                    break;
                }
                done += got;
            }
            close(fd);
        }
    };
    const int nt = threads < 1 ? 1 : (threads > n ? n : (threads > 64 ? 64 : threads));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (failed.load()) { cgs_set_error("file %s: %s", WRITE ? "write" : "read", err.c_str()); return CGS_ERR_ARG; }
    return CGS_OK;
}

}  // namespace

extern "C" int cgs_pread_ranges(int n, const char *const *paths, const int64_t *file_off, const int64_t *nbytes, void *const *dst,
                                int threads) {
    return run_ranges<false>(n, paths, file_off, nbytes, dst, threads);
}

extern "C" int cgs_pwrite_ranges(int n, const char *const *paths, const int64_t *file_off, const int64_t *nbytes,
                                 const void *const *src, int threads) {
    return run_ranges<true>(n, paths, file_off, nbytes, (void *const *)src, threads);
}
